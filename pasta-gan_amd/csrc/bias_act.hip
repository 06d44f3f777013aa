// bias_act for gfx950: y = clamp(act(x + b) * gain) and its first / second derivatives,
// one pass over HBM with 16-byte accesses.
//
// Semantics follow torch_utils/ops/bias_act.py:94-123 (forward) and the derivative /
// clamp-mask rules of torch_utils/ops/bias_act.cu:38-146:
//   grad 0: u = x + b;            y = act(u) * gain;                 clamp the value
//   grad 1: x is dy; yy = yref/gain (u = xref + b for swish);
//           y = x * act'(.) * gain;                                  zero where |yref| >= clamp
//   grad 2: x is d_dx; y = x * act''(.) * gain * dy;                 same mask
// act codes 1..9 = linear, relu, lrelu, tanh, sigmoid, elu, selu, softplus, swish
// (torch_utils/ops/bias_act.py:23-33, field cuda_idx).
#include "common.h"

namespace pasta {

struct BiasActParams {
    const void* x; const void* b; const void* xref; const void* yref; const void* dy; void* y;
    int64_t n; int size_b; int64_t step_b;
    float alpha, gain, clamp;
    float* y_amax;              // optional: PASTA_AMAX_PARTS zeroed floats that receive the largest finite |y| (common.h, amax_commit)
};

template <int A, int G, class S>
__device__ __forceinline__ S bias_act_point(S x, S b, S xref, S yref, S dy, S alpha, S gain, S clamp) {
    const S one = (S)1, two = (S)2;
    const S exp_range = (S)80, half_exp_range = (S)40;
    const S selu_scale = (S)1.0507009873554804934193349852946;
    const S selu_alpha = (S)1.6732632423543772848170429916717;
    S y = 0;
    if (G == 0) x += b; else xref += b;
    const S yy = (gain != 0) ? yref / gain : (S)0;

    if (A == 1) {            // linear
        if (G <= 1) y = x;
    } else if (A == 2) {     // relu
        if (G == 0) y = x > 0 ? x : (S)0;
        if (G == 1) y = yy > 0 ? x : (S)0;
    } else if (A == 3) {     // leaky relu
        if (G == 0) y = x > 0 ? x : x * alpha;
        if (G == 1) y = yy > 0 ? x : x * alpha;
    } else if (A == 4) {     // tanh
        if (G == 0) { S c = exp(x), d = one / c; y = x < -exp_range ? -one : x > exp_range ? one : (c - d) / (c + d); }
        if (G == 1) y = x * (one - yy * yy);
        if (G == 2) y = x * (one - yy * yy) * (-two * yy);
    } else if (A == 5) {     // sigmoid
        if (G == 0) y = x < -exp_range ? (S)0 : one / (exp(-x) + one);
        if (G == 1) y = x * yy * (one - yy);
        if (G == 2) y = x * yy * (one - yy) * (one - two * yy);
    } else if (A == 6) {     // elu
        if (G == 0) y = x >= 0 ? x : exp(x) - one;
        if (G == 1) y = yy >= 0 ? x : x * (yy + one);
        if (G == 2) y = yy >= 0 ? (S)0 : x * (yy + one);
    } else if (A == 7) {     // selu
        if (G == 0) y = x >= 0 ? selu_scale * x : (selu_scale * selu_alpha) * (exp(x) - one);
        if (G == 1) y = yy >= 0 ? x * selu_scale : x * (yy + selu_scale * selu_alpha);
        if (G == 2) y = yy >= 0 ? (S)0 : x * (yy + selu_scale * selu_alpha);
    } else if (A == 8) {     // softplus
        if (G == 0) y = x > exp_range ? x : log(exp(x) + one);
        if (G == 1) y = x * (one - exp(-yy));
        if (G == 2) { S c = exp(-yy); y = x * c * (one - c); }
    } else if (A == 9) {     // swish
        if (G == 0) {
            y = x < -exp_range ? (S)0 : x / (exp(-x) + one);
        } else {
            S c = exp(xref), d = c + one;
            if (G == 1) y = xref > half_exp_range ? x : x * c * (xref + d) / (d * d);
            else        y = xref > half_exp_range ? (S)0 : x * c * (xref * (two - d) + two * d) / (d * d * d);
            yref = xref < -exp_range ? (S)0 : xref / (exp(-xref) + one) * gain;
        }
    }

    y *= gain * dy;
    if (clamp >= 0) {
        if (G == 0) y = (y > -clamp && y < clamp) ? y : (y >= 0 ? clamp : -clamp);
        else        y = (yref > -clamp && yref < clamp) ? y : (S)0;
    }
    return y;
}

// V elements per thread per step (16 bytes); when V > 1 the host guarantees n % V == 0,
// step_b % V == 0 and 16-byte aligned pointers, so a pack never straddles a bias index.
template <class T, int A, int G, int V>
__global__ __launch_bounds__(256) void bias_act_kernel(BiasActParams p) {
    typedef typename acc_of<T>::type S;
    const S alpha = (S)p.alpha, gain = (S)p.gain, clamp = (S)p.clamp;
    const int64_t nv = p.n / V;
    const Pack<T, V>* xs = (const Pack<T, V>*)p.x;
    const Pack<T, V>* xr = (const Pack<T, V>*)p.xref;
    const Pack<T, V>* yr = (const Pack<T, V>*)p.yref;
    const Pack<T, V>* dys = (const Pack<T, V>*)p.dy;
    Pack<T, V>* ys = (Pack<T, V>*)p.y;
    uint32_t am = 0;
    const AmaxSlot aslot = amax_begin(p.y_amax);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (int64_t)gridDim.x * blockDim.x) {
        Pack<T, V> vx = xs[i], vxr, vyr, vdy, out;
        if (xr) vxr = xr[i];
        if (yr) vyr = yr[i];
        if (dys) vdy = dys[i];
        S b = 0;
        if (p.b) b = ld<T>((const T*)p.b + ((i * V) / p.step_b) % p.size_b);
#pragma unroll
        for (int k = 0; k < V; k++) {
            S r = bias_act_point<A, G, S>(ld<T>(&vx.v[k]), b, xr ? ld<T>(&vxr.v[k]) : (S)0, yr ? ld<T>(&vyr.v[k]) : (S)0,
                                          dys ? ld<T>(&vdy.v[k]) : (S)1, alpha, gain, clamp);
            st<T>(&out.v[k], r);
            if (p.y_amax) amax_take(am, (float)ld<T>(&out.v[k]));
        }
        ys[i] = out;
    }
    amax_commit(am, aslot);
}

// The same on tensors whose bias runs are long (or that have no bias): workgroup (chunk, plane) covers PLANE_CHUNK_PACKS packs of
// one bias run, so the bias is a scalar and no index is divided, and a thread fetches its four packs of every operand BEFORE it
// computes or stores anything.  (The operands are not __restrict__ -- dx may be dy -- so in the loop above every load waits behind
// the previous iteration's store: one 16-byte load in flight per thread, a few KB per CU, against the ~72 KB an HBM round trip
// needs; measured 0.51 of the HBM peak on [16, 64, 256, 256], torch's own element-wise add 0.73.)
constexpr int PLANE_CHUNK_PACKS = 1024;

template <class T, int A, int G, int V>
__global__ __launch_bounds__(256) void bias_act_plane_kernel(BiasActParams p, int plane_packs, int chunks, int nitems) {
    typedef typename acc_of<T>::type S;
    constexpr int U = PLANE_CHUNK_PACKS / 256;
    const S alpha = (S)p.alpha, gain = (S)p.gain, clamp = (S)p.clamp;
    const Pack<T, V>* const xs = (const Pack<T, V>*)p.x;
    const Pack<T, V>* const xr = (const Pack<T, V>*)p.xref;
    const Pack<T, V>* const yr = (const Pack<T, V>*)p.yref;
    const Pack<T, V>* const dys = (const Pack<T, V>*)p.dy;
    Pack<T, V>* const ys = (Pack<T, V>*)p.y;
    uint32_t am = 0;
    const AmaxSlot aslot = amax_begin(p.y_amax);
    struct Item { Pack<T, V> vx[U], vxr[U], vyr[U], vdy[U]; S b; };
    // item = (plane, chunk): its packs j0 + 256 u of the plane, clamped to the plane's last pack (fetched, never stored)
    auto load = [&](int item, Item& it) {
        const int plane = item / chunks, chunk = item - plane * chunks;
        const int64_t base = (int64_t)plane * plane_packs;
        const int j0 = chunk * PLANE_CHUNK_PACKS + threadIdx.x;
        it.b = p.b ? ld<T>((const T*)p.b + plane % p.size_b) : (S)0;
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int64_t j = base + (j0 + 256 * u < plane_packs ? j0 + 256 * u : plane_packs - 1);
            it.vx[u] = xs[j];
            if (xr) it.vxr[u] = xr[j];
            if (yr) it.vyr[u] = yr[j];
            if (dys) it.vdy[u] = dys[j];
        }
    };
    auto process = [&](int item, const Item& it) {
        const int plane = item / chunks, chunk = item - plane * chunks;
        const int64_t base = (int64_t)plane * plane_packs;
        const int j0 = chunk * PLANE_CHUNK_PACKS + threadIdx.x;
#pragma unroll
        for (int u = 0; u < U; u++) {
            if (j0 + 256 * u >= plane_packs) continue;
            Pack<T, V> out;
#pragma unroll
            for (int k = 0; k < V; k++) {
                S r = bias_act_point<A, G, S>(ld<T>(&it.vx[u].v[k]), it.b, xr ? ld<T>(&it.vxr[u].v[k]) : (S)0, yr ? ld<T>(&it.vyr[u].v[k]) : (S)0,
                                              dys ? ld<T>(&it.vdy[u].v[k]) : (S)1, alpha, gain, clamp);
                st<T>(&out.v[k], r);
                if (p.y_amax) amax_take(am, (float)ld<T>(&out.v[k]));
            }
            ys[base + j0 + 256 * u] = out;
        }
    };
    // the workgroup walks items blockIdx.x, + gridDim.x, ...: the fetches of the next item are in flight while this one is computed and
    // stored, and the |max| of everything it wrote leaves in ONE commit (a commit per 16 KB of output cost a fifth of the kernel's time)
    Item ia, ib;
    int item = blockIdx.x;
    if (item < nitems) load(item, ia);
    while (item < nitems) {
        const int n1 = item + gridDim.x;
        if (n1 < nitems) load(n1, ib);
        process(item, ia);
        if (n1 >= nitems) break;
        const int n2 = n1 + gridDim.x;
        if (n2 < nitems) load(n2, ia);
        process(n1, ib);
        item = n2;
    }
    __shared__ uint32_t amred[4];
    amax_commit_block<256>(am, aslot, amred);
}

template <class T, int A, int G>
static void launch(const BiasActParams& p, hipStream_t s) {
    constexpr int V = 16 / sizeof(T);
    auto aligned = [](const void* q) { return ((uintptr_t)q & 15) == 0; };
    const bool vec = p.n % V == 0 && (!p.b || p.step_b % V == 0) && aligned(p.x) && aligned(p.y) && aligned(p.xref) &&
                     aligned(p.yref) && aligned(p.dy);
    if (vec && sizeof(T) <= 4) {
        // runs of the bias (the whole tensor when there is none) as planes of 16-byte packs
        const int64_t run = p.b ? p.step_b : p.n, planes = p.n / run, packs = run / V;
        const int64_t chunks = ceil_div64(packs, PLANE_CHUNK_PACKS), nitems = planes * chunks;
        if (p.n % run == 0 && packs >= 256 && packs <= INT32_MAX / 2 && nitems <= INT32_MAX / 2) {
            const int64_t grid = nitems < 256 * 8 ? nitems : 256 * 8;       // eight workgroups per CU walk the items
            hipLaunchKernelGGL((bias_act_plane_kernel<T, A, G, V>), dim3((unsigned)grid), dim3(256), 0, s, p, (int)packs, (int)chunks, (int)nitems);
            return;
        }
    }
    const int64_t work = vec ? p.n / V : p.n;
    int64_t blocks = ceil_div64(work, 256);
    if (blocks > 256 * 16) blocks = 256 * 16;   // grid-stride beyond 16 workgroups per CU
    if (blocks < 1) blocks = 1;
    if (vec) hipLaunchKernelGGL((bias_act_kernel<T, A, G, V>), dim3((unsigned)blocks), dim3(256), 0, s, p);
    else     hipLaunchKernelGGL((bias_act_kernel<T, A, G, 1>), dim3((unsigned)blocks), dim3(256), 0, s, p);
}

template <class T, int A>
static void launch_grad(const BiasActParams& p, int grad, hipStream_t s) {
    if (grad == 0) launch<T, A, 0>(p, s);
    else if (grad == 1) launch<T, A, 1>(p, s);
    else launch<T, A, 2>(p, s);
}

template <class T>
static int launch_act(const BiasActParams& p, int act, int grad, hipStream_t s) {
    switch (act) {
        case 1: launch_grad<T, 1>(p, grad, s); break;
        case 2: launch_grad<T, 2>(p, grad, s); break;
        case 3: launch_grad<T, 3>(p, grad, s); break;
        case 4: launch_grad<T, 4>(p, grad, s); break;
        case 5: launch_grad<T, 5>(p, grad, s); break;
        case 6: launch_grad<T, 6>(p, grad, s); break;
        case 7: launch_grad<T, 7>(p, grad, s); break;
        case 8: launch_grad<T, 8>(p, grad, s); break;
        case 9: launch_grad<T, 9>(p, grad, s); break;
        default: return fail("bias_act: no kernel for activation code %d", act);
    }
    return launch_status("bias_act");
}

//------------------------------------------------------------------------------------
// Bias gradient: db[c] = sum of dx over every element whose bias index is c.
// Element i belongs to c = (i / step_b) % size_b, i.e. dx is viewed as
// [outer, size_b, step_b].  Stage 1: one workgroup per (c, slice of outer) -> partial;
// stage 2: one wave per c sums the partials in a fixed order (bitwise reproducible).

template <class T>
__global__ __launch_bounds__(256) void bias_grad_partial_kernel(const T* dx, float* work, int64_t outer, int size_b,
                                                                int64_t step_b, int nsplit) {
    const int c = blockIdx.x, sp = blockIdx.y;
    const int64_t o0 = outer * sp / nsplit, o1 = outer * (sp + 1) / nsplit;
    float acc = 0.f;
    for (int64_t o = o0; o < o1; o++) {
        const T* row = dx + (o * size_b + c) * step_b;
        for (int64_t j = threadIdx.x; j < step_b; j += 256) acc += (float)ld<T>(row + j);
    }
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    __shared__ float part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) work[(int64_t)c * nsplit + sp] = part[0] + part[1] + part[2] + part[3];
}

template <class T>
__global__ __launch_bounds__(64) void bias_grad_final_kernel(const float* work, T* db, int nsplit) {
    const int c = blockIdx.x;
    float acc = 0.f;
    for (int j = threadIdx.x; j < nsplit; j += 64) acc += work[(int64_t)c * nsplit + j];
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if (threadIdx.x == 0) st<T>(db + c, (typename acc_of<T>::type)acc);
}

static int bias_grad_nsplit(int64_t n, int size_b, int64_t step_b) {
    const int64_t outer = n / ((int64_t)size_b * step_b);
    int64_t want = 2048 / (size_b > 0 ? size_b : 1);   // ~8 workgroups per CU in total
    if (want < 1) want = 1;
    if (want > outer) want = outer;
    if (want < 1) want = 1;
    return (int)want;
}

//------------------------------------------------------------------------------------
// grad 1 with the bias gradient folded in (linear / relu / lrelu on [outer, size_b, step_b] tensors with long
// step_b runs): dx = dy * act'(yref) * gain under the clamp mask, and per workgroup the sum of its dx values, so that
// db needs no second pass over dx.  Workgroup (plane, chunk) covers up to 1024 packs of one (n, c) plane; partial sums
// are laid out [c][n][chunk] and summed per c in a fixed order by bias_grad_final_kernel.
constexpr int DB_CHUNK_PACKS = 1024;

constexpr int DB_SPAN = 8;          // chunks of one plane per workgroup: one reduction and one |max| commit per 128 KB of dx

template <class T, int A, int V>
__global__ __launch_bounds__(256) void bias_act_grad_db_kernel(BiasActParams p, float* work, int outer, int chunks, int spans) {
    typedef typename acc_of<T>::type S;
    constexpr int U = DB_CHUNK_PACKS / 256;
    const S alpha = (S)p.alpha, gain = (S)p.gain, clamp = (S)p.clamp;
    const int plane = blockIdx.x, span = blockIdx.y;
    const int plane_packs = (int)(p.step_b / V);
    const Pack<T, V>* dys = (const Pack<T, V>*)p.x + (int64_t)plane * plane_packs;
    const Pack<T, V>* yr = p.yref ? (const Pack<T, V>*)p.yref + (int64_t)plane * plane_packs : nullptr;
    Pack<T, V>* dxs = (Pack<T, V>*)p.y + (int64_t)plane * plane_packs;
    float acc = 0.f;
    uint32_t am = 0;
    const AmaxSlot aslot = amax_begin(p.y_amax);
    struct Item { Pack<T, V> vdy[U], vyr[U]; };
    // every fetch of a chunk before its arithmetic and stores (dx may be dy: a load behind a store of the same loop would wait for
    // it), and the next chunk's fetches in flight meanwhile
    auto load = [&](int chunk, Item& it) {
        const int jb = chunk * DB_CHUNK_PACKS + threadIdx.x;
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int j = jb + 256 * u < plane_packs ? jb + 256 * u : plane_packs - 1;
            it.vdy[u] = dys[j];
            if (yr) it.vyr[u] = yr[j];
        }
    };
    auto process = [&](int chunk, const Item& it) {
        const int jb = chunk * DB_CHUNK_PACKS + threadIdx.x;
#pragma unroll
        for (int u = 0; u < U; u++) {
            if (jb + 256 * u >= plane_packs) continue;
            Pack<T, V> out;
#pragma unroll
            for (int k = 0; k < V; k++) {
                const S r = bias_act_point<A, 1, S>(ld<T>(&it.vdy[u].v[k]), (S)0, (S)0, yr ? ld<T>(&it.vyr[u].v[k]) : (S)0, (S)1, alpha, gain, clamp);
                st<T>(&out.v[k], r);
                acc += (float)ld<T>(&out.v[k]);          // the stored (rounded) value, as a sum over dx would see it
                if (p.y_amax) amax_take(am, (float)ld<T>(&out.v[k]));
            }
            dxs[jb + 256 * u] = out;
        }
    };
    const int c0 = span * DB_SPAN, c1 = min(chunks, c0 + DB_SPAN);
    Item ia, ib;
    int chunk = c0;
    load(chunk, ia);
    while (chunk < c1) {
        if (chunk + 1 < c1) load(chunk + 1, ib);
        process(chunk, ia);
        if (chunk + 1 >= c1) break;
        if (chunk + 2 < c1) load(chunk + 2, ia);
        process(chunk + 1, ib);
        chunk += 2;
    }
    __shared__ uint32_t amred[4];
    amax_commit_block<256>(am, aslot, amred);
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    __shared__ float part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        const int n = plane / p.size_b, c = plane - n * p.size_b;
        work[((int64_t)c * outer + n) * spans + span] = part[0] + part[1] + part[2] + part[3];
    }
}

// Shapes the fused kernel takes; 0 chunks = not supported (callers use pasta_bias_act + pasta_bias_grad).
static int grad_db_chunks(int dtype, int64_t n, int size_b, int64_t step_b, int act) {
    if (dtype != PASTA_F32 && dtype != PASTA_F16 && dtype != PASTA_BF16) return 0;
    const int V = dtype == PASTA_F32 ? 4 : 8;
    if (act < 1 || act > 3 || n <= 0 || size_b <= 0 || step_b <= 0 || n % ((int64_t)size_b * step_b) != 0) return 0;
    if (step_b % V != 0 || step_b / V < 256 || step_b / V > INT32_MAX / 2 || n / step_b > INT32_MAX) return 0;
    const int64_t chunks = (step_b / V + DB_CHUNK_PACKS - 1) / DB_CHUNK_PACKS;
    return chunks <= 65535 ? (int)chunks : 0;
}

template <class T, int V>
static int launch_grad_db(const BiasActParams& p, float* work, void* db, int act, int outer, int chunks, hipStream_t s) {
    const int spans = (chunks + DB_SPAN - 1) / DB_SPAN;          // partial sums [c][n][span]: the first outer * size_b * spans floats of `work`
    dim3 grid((unsigned)(outer * p.size_b), (unsigned)spans);
    switch (act) {
        case 1: hipLaunchKernelGGL((bias_act_grad_db_kernel<T, 1, V>), grid, dim3(256), 0, s, p, work, outer, chunks, spans); break;
        case 2: hipLaunchKernelGGL((bias_act_grad_db_kernel<T, 2, V>), grid, dim3(256), 0, s, p, work, outer, chunks, spans); break;
        default: hipLaunchKernelGGL((bias_act_grad_db_kernel<T, 3, V>), grid, dim3(256), 0, s, p, work, outer, chunks, spans); break;
    }
    hipLaunchKernelGGL((bias_grad_final_kernel<T>), dim3(p.size_b), dim3(64), 0, s, work, (T*)db, outer * spans);
    return launch_status("bias_act_grad_db");
}

}  // namespace pasta

extern "C" int64_t pasta_bias_act_grad_db_workspace(int dtype, int64_t n, int size_b, int64_t step_b, int act) {
    const int chunks = pasta::grad_db_chunks(dtype, n, size_b, step_b, act);
    return chunks ? (n / step_b) * chunks * (int64_t)sizeof(float) : 0;
}

extern "C" int pasta_bias_act_grad_db(const void* dy, const void* yref, void* dx, void* db, float* work, int dtype, int64_t n,
                                      int size_b, int64_t step_b, int act, float alpha, float gain, float clamp, void* stream, float* dx_amax) {
    using namespace pasta;
    const int chunks = grad_db_chunks(dtype, n, size_b, step_b, act);
    PASTA_CHECK(chunks > 0, "bias_act_grad_db: unsupported case (dtype %d, n %lld, size_b %d, step_b %lld, act %d)", dtype,
                (long long)n, size_b, (long long)step_b, act);
    PASTA_CHECK(dy && dx && db && work, "bias_act_grad_db: null pointer");
    PASTA_CHECK(yref || (act == 1 && clamp < 0), "bias_act_grad_db: yref is required for this activation / clamp");
    PASTA_CHECK((((uintptr_t)dy | (uintptr_t)dx | (uintptr_t)yref) & 15) == 0, "bias_act_grad_db: pointers must be 16-byte aligned");
    BiasActParams p;
    p.x = dy; p.b = nullptr; p.xref = nullptr; p.yref = yref; p.dy = nullptr; p.y = dx;
    p.n = n; p.size_b = size_b; p.step_b = step_b;
    p.alpha = alpha; p.gain = gain; p.clamp = clamp; p.y_amax = dx_amax;
    const int outer = (int)(n / ((int64_t)size_b * step_b));
    hipStream_t s = (hipStream_t)stream;
    if (dtype == PASTA_F32) return launch_grad_db<float, 4>(p, work, db, act, outer, chunks, s);
    if (dtype == PASTA_BF16) return launch_grad_db<__bf16, 8>(p, work, db, act, outer, chunks, s);
    return launch_grad_db<__half, 8>(p, work, db, act, outer, chunks, s);
}

extern "C" int pasta_bias_act(const void* x, const void* b, const void* xref, const void* yref, const void* dy, void* y,
                              int dtype, int64_t n, int size_b, int64_t step_b, int grad, int act, float alpha,
                              float gain, float clamp, void* stream, float* y_amax) {
    using namespace pasta;
    PASTA_CHECK(n >= 0, "bias_act: negative element count");
    if (n == 0) return 0;
    PASTA_CHECK(x && y, "bias_act: null pointer");
    PASTA_CHECK(n <= INT32_MAX, "bias_act: x is too large");
    PASTA_CHECK(grad >= 0 && grad <= 2, "bias_act: grad must be 0, 1 or 2");
    PASTA_CHECK(!b || (size_b >= 1 && step_b >= 1), "bias_act: b has wrong number of elements");
    BiasActParams p;
    p.x = x; p.b = b; p.xref = xref; p.yref = yref; p.dy = dy; p.y = y;
    p.n = n; p.size_b = b ? size_b : 1; p.step_b = b ? step_b : 1;
    p.alpha = alpha; p.gain = gain; p.clamp = clamp; p.y_amax = y_amax;
    hipStream_t s = (hipStream_t)stream;
    switch (dtype) {
        case PASTA_F32: return launch_act<float>(p, act, grad, s);
        case PASTA_F16: return launch_act<__half>(p, act, grad, s);
        case PASTA_BF16: return launch_act<__bf16>(p, act, grad, s);
        case PASTA_F64: return launch_act<double>(p, act, grad, s);
        default: return fail("bias_act: unsupported dtype code %d", dtype);
    }
}

extern "C" int64_t pasta_bias_grad_workspace(int64_t n, int size_b, int64_t step_b) {
    if (n <= 0 || size_b <= 0 || step_b <= 0) return 0;
    return (int64_t)size_b * pasta::bias_grad_nsplit(n, size_b, step_b) * (int64_t)sizeof(float);
}

extern "C" int pasta_bias_grad(const void* dx, void* db, float* work, int dtype, int64_t n, int size_b, int64_t step_b,
                               void* stream) {
    using namespace pasta;
    PASTA_CHECK(dx && db && work, "bias_grad: null pointer");
    PASTA_CHECK(size_b >= 1 && step_b >= 1 && n >= 1 && n % ((int64_t)size_b * step_b) == 0,
                "bias_grad: n=%lld is not a multiple of size_b*step_b", (long long)n);
    const int64_t outer = n / ((int64_t)size_b * step_b);
    const int nsplit = bias_grad_nsplit(n, size_b, step_b);
    hipStream_t s = (hipStream_t)stream;
    dim3 grid(size_b, nsplit);
    switch (dtype) {
        case PASTA_F32:
            hipLaunchKernelGGL((bias_grad_partial_kernel<float>), grid, dim3(256), 0, s, (const float*)dx, work, outer, size_b, step_b, nsplit);
            hipLaunchKernelGGL((bias_grad_final_kernel<float>), dim3(size_b), dim3(64), 0, s, work, (float*)db, nsplit);
            break;
        case PASTA_F16:
            hipLaunchKernelGGL((bias_grad_partial_kernel<__half>), grid, dim3(256), 0, s, (const __half*)dx, work, outer, size_b, step_b, nsplit);
            hipLaunchKernelGGL((bias_grad_final_kernel<__half>), dim3(size_b), dim3(64), 0, s, work, (__half*)db, nsplit);
            break;
        case PASTA_BF16:
            hipLaunchKernelGGL((bias_grad_partial_kernel<__bf16>), grid, dim3(256), 0, s, (const __bf16*)dx, work, outer, size_b, step_b, nsplit);
            hipLaunchKernelGGL((bias_grad_final_kernel<__bf16>), dim3(size_b), dim3(64), 0, s, work, (__bf16*)db, nsplit);
            break;
        case PASTA_F64:
            hipLaunchKernelGGL((bias_grad_partial_kernel<double>), grid, dim3(256), 0, s, (const double*)dx, work, outer, size_b, step_b, nsplit);
            hipLaunchKernelGGL((bias_grad_final_kernel<double>), dim3(size_b), dim3(64), 0, s, work, (double*)db, nsplit);
            break;
        default: return fail("bias_grad: unsupported dtype code %d", dtype);
    }
    return launch_status("bias_grad");
}
