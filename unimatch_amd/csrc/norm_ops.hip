// Fused InstanceNorm2d (affine = False) + ReLU (+ shortcut add + ReLU) for the CNN encoder.      gfx950 / wave64
//
//   t = (x - mean_plane) / sqrt(var_plane + eps);   if relu: t = max(t, 0);   if shortcut: t = max(t + shortcut, 0)
//
// Replaces the ATen chain batch_norm_collect_statistics -> batch_norm_transform_input -> clamp_min (-> add ->
// clamp_min) that the reference's ResidualBlock (unimatch/backbone.py:31-36) turns into on the GPU: 6 reads + 4
// writes of the activation per block tail become (at most) 2 HBM reads + 1 write.  HBM/L2-bound streaming kernel:
// one workgroup per (image, channel) plane, float4 accesses, two-pass statistics (mean, then centred variance --
// the plane is L2 resident for the second and third sweep), wave shuffles + one LDS hop for the block reduction.
// Outside SURVEY section 8's hot path; kept because the encoder's element-wise tail had become 18 % of the step.
#include "common.h"
#include "timing.h"

__device__ __forceinline__ float block_sum(float v, float* red) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    const int wave = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[wave] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(256) void instance_norm_kernel(const float* __restrict__ x,
                                                            const float* __restrict__ shortcut,
                                                            float* __restrict__ y, int hw, float eps, int relu) {
    __shared__ float red[4];
    const long base = (long)blockIdx.x * hw;
    const float* xp = x + base;
    const int n4 = hw >> 2, tid = threadIdx.x;
    float s = 0.f;
    for (int i = tid; i < n4; i += 256) {
        const f32x4 v = reinterpret_cast<const f32x4*>(xp)[i];
        s += (v[0] + v[1]) + (v[2] + v[3]);
    }
    for (int i = 4 * n4 + tid; i < hw; i += 256) s += xp[i];
    const float mean = block_sum(s, red) / (float)hw;
    float q = 0.f;
    for (int i = tid; i < n4; i += 256) {
        const f32x4 v = reinterpret_cast<const f32x4*>(xp)[i];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float d = v[j] - mean;
            q = __builtin_fmaf(d, d, q);
        }
    }
    for (int i = 4 * n4 + tid; i < hw; i += 256) {
        const float d = xp[i] - mean;
        q = __builtin_fmaf(d, d, q);
    }
    const float rstd = 1.0f / sqrtf(block_sum(q, red) / (float)hw + eps);
    const float* sp = shortcut ? shortcut + base : nullptr;
    float* yp = y + base;
    for (int i = tid; i < n4; i += 256) {
        f32x4 v = reinterpret_cast<const f32x4*>(xp)[i];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            v[j] = (v[j] - mean) * rstd;
            if (relu) v[j] = fmaxf(v[j], 0.f);
        }
        if (sp) {
            const f32x4 r = reinterpret_cast<const f32x4*>(sp)[i];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j] + r[j], 0.f);
        }
        reinterpret_cast<f32x4*>(yp)[i] = v;
    }
    for (int i = 4 * n4 + tid; i < hw; i += 256) {
        float v = (xp[i] - mean) * rstd;
        if (relu) v = fmaxf(v, 0.f);
        if (sp) v = fmaxf(v + sp[i], 0.f);
        yp[i] = v;
    }
}

// Register-resident variant: the whole (image, channel) plane lives in the workgroup's registers (up to 1024
// threads x EPT float4), so the activation is read from HBM exactly once -- the planes of the early encoder stages
// (393 KB each, ~1000 of them in flight) do not survive in L2 between the sweeps of the streaming kernel above.
template <int EPT>
__global__ __launch_bounds__(1024) void instance_norm_reg_kernel(const float* __restrict__ x,
                                                                 const float* __restrict__ shortcut,
                                                                 float* __restrict__ y, int hw, float eps, int relu) {
    __shared__ float red[2][16];
    const long base = (long)blockIdx.x * hw;
    const f32x4* xp = reinterpret_cast<const f32x4*>(x + base);
    const int n4 = hw >> 2, tid = threadIdx.x, nthr = blockDim.x;
    const int lane = tid & 63, wave = tid >> 6, nwave = nthr >> 6;
    f32x4 v[EPT];
    float s = 0.f;
    {   // one 32-bit element offset walked by nthr per step (1024 threads = 128 VGPRs per lane, 96 of them hold the plane: separately
        // computed 64-bit addresses of the 24 loads spilled 92 bytes per lane)
        unsigned off = (unsigned)tid;
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            v[e] = off < (unsigned)n4 ? xp[off] : f32x4{0.f, 0.f, 0.f, 0.f};
            off += (unsigned)nthr;
            asm volatile("" : "+v"(off));
        }
    }
#pragma unroll
    for (int e = 0; e < EPT; ++e) s += (v[e][0] + v[e][1]) + (v[e][2] + v[e][3]);
    auto block_total = [&](float val, int slot) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) val += __shfl_xor(val, off);
        if (lane == 0) red[slot][wave] = val;
        __syncthreads();
        float t = 0.f;
        for (int w2 = 0; w2 < nwave; ++w2) t += red[slot][w2];
        return t;
    };
    const float mean = block_total(s, 0) / (float)hw;
    float q = 0.f;
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const int i = tid + e * nthr;
        if (i < n4) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float d = v[e][j] - mean;
                q = __builtin_fmaf(d, d, q);
            }
        }
    }
    const float rstd = 1.0f / sqrtf(block_total(q, 1) / (float)hw + eps);
    const f32x4* sp = shortcut ? reinterpret_cast<const f32x4*>(shortcut + base) : nullptr;
    f32x4* yp = reinterpret_cast<f32x4*>(y + base);
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const int i = tid + e * nthr;
        if (i < n4) {
            f32x4 o = v[e];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                o[j] = (o[j] - mean) * rstd;
                if (relu) o[j] = fmaxf(o[j], 0.f);
            }
            if (sp) {
                const f32x4 r = sp[i];
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = fmaxf(o[j] + r[j], 0.f);
            }
            yp[i] = o;
        }
    }
}

extern void um_set_error(const char* fmt, ...);

extern "C" int um_instance_norm_fwd(const float* x, const float* shortcut, float* y, long planes, int hw, float eps,
                                    int relu, void* stream) {
    if (!x || !y || planes <= 0 || planes > 0x7fffffffL || hw <= 0) {
        um_set_error("um_instance_norm_fwd: bad argument (planes=%ld hw=%d)", planes, hw);
        return -1;
    }
    if ((((uintptr_t)x | (uintptr_t)y | (uintptr_t)shortcut) & 15) != 0 || (hw & 3) != 0) {
        // float4 path needs 16-byte aligned planes; fall back is not provided on purpose (all encoder maps qualify)
        um_set_error("um_instance_norm_fwd: planes must be 16-byte aligned and hw a multiple of 4 (hw=%d)", hw);
        return -4;
    }
    ScopedKernelTimer timer(UM_K_INSTANCE_NORM, (hipStream_t)stream);
    constexpr int EPT = 24;
    const int n4 = hw >> 2;
    if (n4 <= EPT * 1024) {
        int threads = ((n4 + EPT - 1) / EPT + 63) / 64 * 64;
        if (threads < 64) threads = 64;
        hipLaunchKernelGGL((instance_norm_reg_kernel<EPT>), dim3((unsigned)planes), dim3(threads), 0, (hipStream_t)stream, x,
                           shortcut, y, hw, eps, relu);
    } else {
        hipLaunchKernelGGL(instance_norm_kernel, dim3((unsigned)planes), dim3(256), 0, (hipStream_t)stream, x, shortcut, y,
                           hw, eps, relu);
    }
    return (int)hipGetLastError();
}
