// Error plumbing + weight packing (fp32 parameter layout -> MFMA operand layout).
#include <stdarg.h>

#include "common.h"

static thread_local char g_err[512] = "";

void hipseg_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" const char* hipseg_last_error(void) { return g_err; }

#include <map>
#include <mutex>
#include <set>
#include <utility>

int hs_set_max_lds(const void* kernel, size_t bytes) {
    static std::mutex mu;
    static std::set<std::pair<int, const void*>> done;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = -1;
    std::lock_guard<std::mutex> lock(mu);
    if (done.count({dev, kernel})) return HIPSEG_OK;
    const hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) {
        hipseg_set_error("hipFuncSetAttribute(MaxDynamicSharedMemorySize = %zu): %s", bytes, hipGetErrorString(e));
        return HIPSEG_EHIP;
    }
    done.insert({dev, kernel});
    return HIPSEG_OK;
}

int device_cus() {
    static std::mutex mu;
    static std::map<int, int> cus;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    std::lock_guard<std::mutex> lock(mu);
    auto it = cus.find(dev);
    if (it != cus.end()) return it->second;
    hipDeviceProp_t prop;
    const int n = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
    cus[dev] = n;
    return n;
}
extern "C" int hipseg_abi_version(void) { return 1; }

namespace {

// packed layout [tap][Kp/G][Np][G]
template <typename T>
__global__ void pack_conv_kernel(const float* __restrict__ w, T* __restrict__ wp, int Cout, int Cin, int ks,
                                 int transpose, int Kp, int Np, int G, long total) {
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int g = (int)(idx % G);
        const int n = (int)((idx / G) % Np);
        const int kg = (int)((idx / ((long)G * Np)) % (Kp / G));
        const int tap = (int)(idx / ((long)Kp * Np));
        const int k = kg * G + g;
        const int ky = tap / ks, kx = tap % ks;
        float v = 0.f;
        if (!transpose) {  // K = Cin, N = Cout
            if (k < Cin && n < Cout) v = w[(((long)n * Cin + k) * ks + ky) * ks + kx];
        } else {  // K = Cout, N = Cin, taps flipped
            if (k < Cout && n < Cin) v = w[(((long)k * Cin + n) * ks + (ks - 1 - ky)) * ks + (ks - 1 - kx)];
        }
        wp[idx] = (T)v;
    }
}

// forward AND data-gradient operand of one Conv2d weight in a single launch (one launch per layer and step)
template <typename T>
__global__ void pack_conv_both_kernel(const float* __restrict__ w, T* __restrict__ wp, T* __restrict__ wpt, int Cout,
                                      int Cin, int ks, int Kp0, int Np0, int Kp1, int Np1, int G, long total0,
                                      long total1) {
    const long total = total0 > total1 ? total0 : total1;
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        if (idx < total0) {  // K = Cin, N = Cout
            const int g = (int)(idx % G);
            const int n = (int)((idx / G) % Np0);
            const int kg = (int)((idx / ((long)G * Np0)) % (Kp0 / G));
            const int tap = (int)(idx / ((long)Kp0 * Np0));
            const int k = kg * G + g, ky = tap / ks, kx = tap % ks;
            float v = 0.f;
            if (k < Cin && n < Cout) v = w[(((long)n * Cin + k) * ks + ky) * ks + kx];
            wp[idx] = (T)v;
        }
        if (idx < total1) {  // K = Cout, N = Cin, taps flipped
            const int g = (int)(idx % G);
            const int n = (int)((idx / G) % Np1);
            const int kg = (int)((idx / ((long)G * Np1)) % (Kp1 / G));
            const int tap = (int)(idx / ((long)Kp1 * Np1));
            const int k = kg * G + g, ky = tap / ks, kx = tap % ks;
            float v = 0.f;
            if (k < Cout && n < Cin) v = w[(((long)k * Cin + n) * ks + (ks - 1 - ky)) * ks + (ks - 1 - kx)];
            wpt[idx] = (T)v;
        }
    }
}

// ConvTranspose2d weight (Cin, Cout, 2, 2)
template <typename T>
__global__ void pack_convT_kernel(const float* __restrict__ w, T* __restrict__ wp, int Cin, int Cout, int transpose,
                                  int Kp, int Np, int G, long total) {
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int g = (int)(idx % G);
        const int n = (int)((idx / G) % Np);
        const int kg = (int)((idx / ((long)G * Np)) % (Kp / G));
        const int tap = (int)(idx / ((long)Kp * Np));
        const int k = kg * G + g;
        float v = 0.f;
        if (!transpose) {  // forward: 1 tap, K = Cin, N = 4*Cout, n = ab*Cout + co
            if (k < Cin && n < 4 * Cout) {
                const int ab = n / Cout, co = n % Cout;
                v = w[((long)k * Cout + co) * 4 + ab];
            }
        } else {  // data gradient: tap = ab, K = Cout, N = Cin
            if (k < Cout && n < Cin) v = w[((long)n * Cout + k) * 4 + tap];
        }
        wp[idx] = (T)v;
    }
}

// One launch for ALL conv / ConvT weights of a model (blockIdx.y = descriptor): the per-layer pack launches are
// ~5-7 us latency-bound helpers, ~24 of them per U-Net step.
struct PackDesc {
    const float* w;
    void* wp;
    void* wpt;
    int kind;  // 0: Conv2d (Cout,Cin,ks,ks) -> forward + data-gradient operands; 1: ConvTranspose2d (Cin,Cout,2,2)
    int Cout, Cin, ks;
    int Kp0, Np0, Kp1, Np1;
    long total0, total1;
};

// thread = one G-element run (tap, k-group, n) of one operand: G gathered fp32 reads, ONE G*sizeof(T)-byte store
// (16 B for bf16).  n is the fastest index, so a wave writes 1 KiB contiguous; all index math is 32-bit.
template <typename T, int G>
__global__ __launch_bounds__(256) void pack_batch_kernel(const PackDesc* __restrict__ descs) {
    const PackDesc d = descs[blockIdx.y];
    const float* __restrict__ w = d.w;
    const int ks = d.ks, Cout = d.Cout, Cin = d.Cin, kk = ks * ks;
    if constexpr (G == 8) {
        // 3x3 weights with unpadded operands (every conv of the U-Nets): LDS-staged tile transposer.  The gathered path
        // below reads each fp32 element with its own 4-byte load at a 36-byte (or Cin*36-byte) stride -- one 64-byte
        // sector per element, ~1 GB of L2 sector traffic for 31 MB of weights.  Here a block reads 8 output-channel rows
        // x KT input channels x 9 taps as 8 contiguous runs, and both operands are cut out of the LDS copy: forward
        // wp[tap][k/8][n][8 k] and data-gradient wpt[8 - tap][n/8][k][8 n]; LDS rows of 577 floats keep both read
        // patterns (stride 9 across k, stride 577 across n) bank-conflict-free.
        if (d.kind == 0 && ks == 3 && Cin % 32 == 0 && Cout % 8 == 0 && d.Kp0 == Cin && d.Kp1 == Cout && d.Np0 == Cout &&
            d.Np1 == Cin) {
            __shared__ float tile[8][64 * 9 + 1];
            const int KT = Cin % 64 == 0 ? 64 : 32, nkt = Cin / KT, ntl = nkt * (Cout / 8), row = KT * 9;
            typedef T vec8 __attribute__((ext_vector_type(8)));
            for (int tl = blockIdx.x; tl < ntl; tl += gridDim.x) {
                const int kt = tl % nkt, ng = tl / nkt, n0 = ng * 8, k0 = kt * KT;
                for (int i = threadIdx.x; i < 8 * row; i += 256) {
                    const int r = i / row, c = i - r * row;
                    tile[r][c] = w[((size_t)(n0 + r) * Cin + k0) * 9 + c];
                }
                __syncthreads();
                for (int i = threadIdx.x; i < 9 * KT; i += 256) {  // forward: (tap, k-octet, n)
                    const int n = i & 7, kgl = (i >> 3) % (KT / 8), t = i / KT;
                    vec8 o;
#pragma unroll
                    for (int g = 0; g < 8; ++g) o[g] = (T)tile[n][(kgl * 8 + g) * 9 + t];
                    T* dst = reinterpret_cast<T*>(d.wp) + (((size_t)t * (Cin / 8) + k0 / 8 + kgl) * d.Np0 + n0 + n) * 8;
                    *reinterpret_cast<vec8*>(dst) = o;
                }
                for (int i = threadIdx.x; i < 9 * KT; i += 256) {  // data gradient: (tap, k), taps flipped
                    const int k = i % KT, t = i / KT;
                    vec8 o;
#pragma unroll
                    for (int g = 0; g < 8; ++g) o[g] = (T)tile[g][k * 9 + t];
                    T* dst = reinterpret_cast<T*>(d.wpt) + (((size_t)(8 - t) * (Cout / 8) + ng) * d.Np1 + k0 + k) * 8;
                    *reinterpret_cast<vec8*>(dst) = o;
                }
                __syncthreads();
            }
            return;
        }
    }
    const int nv0 = (int)(d.total0 / G), nv1 = (int)(d.total1 / G);
    const int stride = gridDim.x * blockDim.x;
    typedef T vecT __attribute__((ext_vector_type(G)));
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < nv0 + nv1; idx += stride) {
        const bool second = idx >= nv0;
        const int v = second ? idx - nv0 : idx;
        const int Np = second ? d.Np1 : d.Np0, KG = (second ? d.Kp1 : d.Kp0) / G;
        const int n = v % Np, kg = (v / Np) % KG, tap = v / (Np * KG);
        float val[G];
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int k = kg * G + g;
            float x = 0.f;
            if (d.kind == 0) {
                const int ky = tap / ks, kx = tap - ky * ks;
                if (!second) {  // forward operand: K = Cin, N = Cout
                    if (k < Cin && n < Cout) x = w[((long)n * Cin + k) * kk + ky * ks + kx];
                } else {        // data-gradient operand: K = Cout, N = Cin, taps flipped
                    if (k < Cout && n < Cin) x = w[((long)k * Cin + n) * kk + (ks - 1 - ky) * ks + (ks - 1 - kx)];
                }
            } else if (!second) {  // ConvT forward: 1 tap, K = Cin, N = 4*Cout, n = ab*Cout + co
                if (k < Cin && n < 4 * Cout) {
                    const int ab = n / Cout, co = n - ab * Cout;
                    x = w[((long)k * Cout + co) * 4 + ab];
                }
            } else {               // ConvT data gradient: tap = ab, K = Cout, N = Cin
                if (k < Cout && n < Cin) x = w[((long)n * Cout + k) * 4 + tap];
            }
            val[g] = x;
        }
        vecT o;
#pragma unroll
        for (int g = 0; g < G; ++g) o[g] = (T)val[g];
        T* dst = reinterpret_cast<T*>(second ? d.wpt : d.wp) + (size_t)v * G;
        *reinterpret_cast<vecT*>(dst) = o;
    }
}

inline int grid_for(long total) {
    long g = (total + 255) / 256;
    return (int)(g > 2048 ? 2048 : (g < 1 ? 1 : g));
}

}  // namespace

extern "C" int hipseg_pack_conv_weight(const float* w, void* wp, int dtype, int Cout, int Cin, int ksize,
                                       int transpose, hipseg_stream_t stream) {
    HS_REQUIRE(w && wp && Cout > 0 && Cin > 0 && (ksize == 1 || ksize == 3), "pack_conv_weight: bad arguments");
    HS_REQUIRE(dtype == HIPSEG_F32 || dtype == HIPSEG_BF16, "pack_conv_weight: bad dtype");
    const int K = transpose ? Cout : Cin, N = transpose ? Cin : Cout;
    const int Kp = hipseg_kpad(K, dtype), Np = hipseg_npad(N), G = dtype == HIPSEG_BF16 ? 8 : 1;
    const long total = (long)ksize * ksize * Kp * Np;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (dtype == HIPSEG_BF16)
        hipLaunchKernelGGL(pack_conv_kernel<bf16>, dim3(grid_for(total)), dim3(256), 0, s, w, (bf16*)wp, Cout, Cin,
                           ksize, transpose, Kp, Np, G, total);
    else
        hipLaunchKernelGGL(pack_conv_kernel<float>, dim3(grid_for(total)), dim3(256), 0, s, w, (float*)wp, Cout, Cin,
                           ksize, transpose, Kp, Np, G, total);
    HS_LAUNCH_CHECK("pack_conv_weight");
    return HIPSEG_OK;
}

extern "C" int hipseg_pack_conv_weight_both(const float* w, void* wp, void* wpt, int dtype, int Cout, int Cin, int ksize,
                                            hipseg_stream_t stream) {
    HS_REQUIRE(w && wp && wpt && Cout > 0 && Cin > 0 && (ksize == 1 || ksize == 3), "pack_conv_weight_both: bad arguments");
    HS_REQUIRE(dtype == HIPSEG_F32 || dtype == HIPSEG_BF16, "pack_conv_weight_both: bad dtype");
    const int G = dtype == HIPSEG_BF16 ? 8 : 1;
    const int Kp0 = hipseg_kpad(Cin, dtype), Np0 = hipseg_npad(Cout), Kp1 = hipseg_kpad(Cout, dtype), Np1 = hipseg_npad(Cin);
    const long total0 = (long)ksize * ksize * Kp0 * Np0, total1 = (long)ksize * ksize * Kp1 * Np1;
    const long total = total0 > total1 ? total0 : total1;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (dtype == HIPSEG_BF16)
        hipLaunchKernelGGL(pack_conv_both_kernel<bf16>, dim3(grid_for(total)), dim3(256), 0, s, w, (bf16*)wp, (bf16*)wpt,
                           Cout, Cin, ksize, Kp0, Np0, Kp1, Np1, G, total0, total1);
    else
        hipLaunchKernelGGL(pack_conv_both_kernel<float>, dim3(grid_for(total)), dim3(256), 0, s, w, (float*)wp,
                           (float*)wpt, Cout, Cin, ksize, Kp0, Np0, Kp1, Np1, G, total0, total1);
    HS_LAUNCH_CHECK("pack_conv_weight_both");
    return HIPSEG_OK;
}

extern "C" int hipseg_pack_convT_weight(const float* w, void* wp, int dtype, int Cin, int Cout, int transpose,
                                        hipseg_stream_t stream) {
    HS_REQUIRE(w && wp && Cout > 0 && Cin > 0, "pack_convT_weight: bad arguments");
    HS_REQUIRE(dtype == HIPSEG_F32 || dtype == HIPSEG_BF16, "pack_convT_weight: bad dtype");
    const int K = transpose ? Cout : Cin, N = transpose ? Cin : 4 * Cout, NT = transpose ? 4 : 1;
    const int Kp = hipseg_kpad(K, dtype), Np = hipseg_npad(N), G = dtype == HIPSEG_BF16 ? 8 : 1;
    const long total = (long)NT * Kp * Np;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (dtype == HIPSEG_BF16)
        hipLaunchKernelGGL(pack_convT_kernel<bf16>, dim3(grid_for(total)), dim3(256), 0, s, w, (bf16*)wp, Cin, Cout,
                           transpose, Kp, Np, G, total);
    else
        hipLaunchKernelGGL(pack_convT_kernel<float>, dim3(grid_for(total)), dim3(256), 0, s, w, (float*)wp, Cin, Cout,
                           transpose, Kp, Np, G, total);
    HS_LAUNCH_CHECK("pack_convT_weight");
    return HIPSEG_OK;
}

extern "C" size_t hipseg_pack_desc_size(void) { return sizeof(PackDesc); }

extern "C" int hipseg_pack_desc_fill(void* host_descs, int index, const float* w, void* wp, void* wpt, int kind,
                                     int dtype, int Cout, int Cin, int ksize) {
    HS_REQUIRE(host_descs && index >= 0 && w && wp && wpt && Cout > 0 && Cin > 0, "pack_desc_fill: bad arguments");
    HS_REQUIRE(dtype == HIPSEG_F32 || dtype == HIPSEG_BF16, "pack_desc_fill: bad dtype");
    HS_REQUIRE((kind == 0 && (ksize == 1 || ksize == 3)) || (kind == 1 && ksize == 2), "pack_desc_fill: bad kind/ksize");
    PackDesc& d = reinterpret_cast<PackDesc*>(host_descs)[index];
    d.w = w;
    d.wp = wp;
    d.wpt = wpt;
    d.kind = kind;
    d.Cout = Cout;
    d.Cin = Cin;
    d.ks = ksize;
    if (kind == 0) {
        d.Kp0 = hipseg_kpad(Cin, dtype);
        d.Np0 = hipseg_npad(Cout);
        d.Kp1 = hipseg_kpad(Cout, dtype);
        d.Np1 = hipseg_npad(Cin);
        d.total0 = (long)ksize * ksize * d.Kp0 * d.Np0;
        d.total1 = (long)ksize * ksize * d.Kp1 * d.Np1;
    } else {
        d.Kp0 = hipseg_kpad(Cin, dtype);
        d.Np0 = hipseg_npad(4 * Cout);
        d.Kp1 = hipseg_kpad(Cout, dtype);
        d.Np1 = hipseg_npad(Cin);
        d.total0 = (long)d.Kp0 * d.Np0;
        d.total1 = (long)4 * d.Kp1 * d.Np1;
    }
    return HIPSEG_OK;
}

extern "C" int hipseg_pack_batch(const void* dev_descs, int n, int dtype, long max_total, hipseg_stream_t stream) {
    HS_REQUIRE(dev_descs && n > 0 && max_total > 0, "pack_batch: bad arguments");
    HS_REQUIRE(dtype == HIPSEG_F32 || dtype == HIPSEG_BF16, "pack_batch: bad dtype");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    // max_total bounds one operand; a descriptor's two operands are walked by one grid-stride loop
    const int G = dtype == HIPSEG_BF16 ? 8 : 1;
    long gx = (2 * max_total / G + 1023) / 1024;  // ~4 runs per thread for the largest weight
    if (gx > 2048) gx = 2048;
    if (gx < 1) gx = 1;
    const dim3 grid((unsigned)gx, (unsigned)n);
    if (dtype == HIPSEG_BF16)
        hipLaunchKernelGGL((pack_batch_kernel<bf16, 8>), grid, dim3(256), 0, s, reinterpret_cast<const PackDesc*>(dev_descs));
    else
        hipLaunchKernelGGL((pack_batch_kernel<float, 1>), grid, dim3(256), 0, s, reinterpret_cast<const PackDesc*>(dev_descs));
    HS_LAUNCH_CHECK("pack_batch");
    return HIPSEG_OK;
}
