// Stream/event plumbing for the data-parallel step: EXTERNAL event-record nodes inside a captured hipGraph, so that work
// issued eagerly on another stream (RCCL bucket all-reduces) can start when the replayed graph passes a given point.
// (torch.cuda.Event(external=True) is refused by PyTorch on ROCm; the HIP runtime itself implements
// hipEventRecordWithFlags(hipEventRecordExternal).)
#include "common.h"

extern "C" int hipseg_event_create(void** event) {
    HS_REQUIRE(event, "event_create: null output");
    hipEvent_t e;
    hipError_t rc = hipEventCreateWithFlags(&e, hipEventDisableTiming);
    if (rc != hipSuccess) {
        hipseg_set_error("event_create: %s", hipGetErrorString(rc));
        return HIPSEG_EHIP;
    }
    *event = e;
    return HIPSEG_OK;
}

extern "C" int hipseg_event_destroy(void* event) {
    if (event) (void)hipEventDestroy(reinterpret_cast<hipEvent_t>(event));
    return HIPSEG_OK;
}

// Record `event` on `stream`.  While the stream is being captured this adds an EXTERNAL event-record node: every replay
// of the graph records the event when execution reaches that node.  Outside a capture it is a plain record.
extern "C" int hipseg_event_record_external(void* event, hipseg_stream_t stream) {
    HS_REQUIRE(event, "event_record_external: null event");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    hipError_t rc = hipStreamIsCapturing(st, &cs);
    if (rc != hipSuccess) {
        hipseg_set_error("event_record_external: hipStreamIsCapturing: %s", hipGetErrorString(rc));
        return HIPSEG_EHIP;
    }
    if (cs == hipStreamCaptureStatusActive) {
        // hipEventRecordWithFlags(hipEventRecordExternal) is refused on a capturing stream by this runtime (ROCm 7.2:
        // invalid argument), so the event-record node is added to the graph under capture explicitly, behind the
        // stream's current capture dependencies, and becomes the stream's new dependency
        hipGraph_t graph = nullptr;
        const hipGraphNode_t* deps = nullptr;
        size_t ndeps = 0;
        unsigned long long id = 0;
        rc = hipStreamGetCaptureInfo_v2(st, &cs, &id, &graph, &deps, &ndeps);
        hipGraphNode_t node = nullptr;
        if (rc == hipSuccess) rc = hipGraphAddEventRecordNode(&node, graph, deps, ndeps, reinterpret_cast<hipEvent_t>(event));
        if (rc == hipSuccess) rc = hipStreamUpdateCaptureDependencies(st, &node, 1, hipStreamSetCaptureDependencies);
    } else {
        rc = hipEventRecord(reinterpret_cast<hipEvent_t>(event), st);
    }
    if (rc != hipSuccess) {
        hipseg_set_error("event_record_external: record (capturing=%d, event %p, stream %p): %s",
                         (int)(cs == hipStreamCaptureStatusActive), event, (void*)st, hipGetErrorString(rc));
        return HIPSEG_EHIP;
    }
    return HIPSEG_OK;
}

// Make everything issued to `stream` after this call wait for the event's latest record (for an event recorded by a graph
// node: the record of the most recently launched replay).
extern "C" int hipseg_stream_wait_event(hipseg_stream_t stream, void* event) {
    HS_REQUIRE(event, "stream_wait_event: null event");
    hipError_t rc = hipStreamWaitEvent(reinterpret_cast<hipStream_t>(stream), reinterpret_cast<hipEvent_t>(event), 0);
    if (rc != hipSuccess) {
        hipseg_set_error("stream_wait_event: %s", hipGetErrorString(rc));
        return HIPSEG_EHIP;
    }
    return HIPSEG_OK;
}

// ---------------------------------------------------------------------------------------------------
// Gradient-bucket all-reduce (average) over RCCL for hosts that own an RCCL communicator themselves
// (`bucket_allreduce(ptr, count, dtype, comm, stream)`; replaces the NCCL all-reduce torch DDP issues
// per bucket for /root/reference/scripts/train_distributed.py:35).  The Python host of this repo goes through
// torch.distributed instead (its process group owns the communicator and does not hand it out): hipseg/ddp.py.
// RCCL is bound at the first call (dlopen of the librccl the process already has, else the system's), so libhipseg.so
// itself carries no link-time dependency on it.
#include <dlfcn.h>

namespace {
typedef int (*nccl_allreduce_fn)(const void*, void*, size_t, int, int, void*, hipStream_t);
nccl_allreduce_fn rccl_allreduce() {
    static nccl_allreduce_fn fn = []() -> nccl_allreduce_fn {
        void* sym = dlsym(RTLD_DEFAULT, "ncclAllReduce");  // (a process that imported torch has its RCCL loaded already)
        if (!sym) {
            for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so"}) {
                if (void* h = dlopen(name, RTLD_NOW | RTLD_GLOBAL)) {
                    sym = dlsym(h, "ncclAllReduce");
                    if (sym) break;
                }
            }
        }
        return reinterpret_cast<nccl_allreduce_fn>(sym);
    }();
    return fn;
}
}  // namespace

extern "C" int hipseg_bucket_allreduce(void* bucket, size_t count, int dtype, void* comm, hipseg_stream_t stream) {
    HS_REQUIRE(bucket && count > 0 && comm, "bucket_allreduce: null bucket / communicator or empty bucket");
    HS_REQUIRE(dtype == HIPSEG_F32, "bucket_allreduce: gradient buckets are fp32 (dtype %d)", dtype);
    nccl_allreduce_fn fn = rccl_allreduce();
    HS_REQUIRE(fn, "bucket_allreduce: no RCCL in this process (ncclAllReduce not found)");
    constexpr int kNcclFloat32 = 7, kNcclAvg = 4;  // ncclDataType_t / ncclRedOp_t of rccl.h (checked in the GPU test)
    const int rc = fn(bucket, bucket, count, kNcclFloat32, kNcclAvg, comm, reinterpret_cast<hipStream_t>(stream));
    if (rc != 0) {
        hipseg_set_error("bucket_allreduce: ncclAllReduce returned %d", rc);
        return HIPSEG_EHIP;
    }
    return HIPSEG_OK;
}
