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
    hipLaunchKernelGGL(instance_norm_kernel, dim3((unsigned)planes), dim3(256), 0, (hipStream_t)stream, x, shortcut, y, hw,
                       eps, relu);
    return (int)hipGetLastError();
}
