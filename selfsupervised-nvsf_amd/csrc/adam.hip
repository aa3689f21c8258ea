// Adam update of the training step (gfx950): the optimiser the reference builds at main_nvsf.py:350-352
// (torch.optim.Adam, betas 0.9 / 0.99, eps 1e-15, no weight decay, no amsgrad) under its GradScaler (trainer.py:119, 1332-1334),
// as ONE streaming pass per parameter tensor (16 B read + 12 B written per parameter) instead of the seven multi-tensor
// passes of torch's default implementation, and without the scaler's device->host read of the overflow flag:
//   nvsf_adam_prepare  one thread: skip = (found_inf != 0); step += !skip; bias corrections of that step in double
//   nvsf_adam_update   g = grad / grad_scale;  m += (1 - b1)(g - m);  v = b2 v + (1 - b2) g g;
//                      p -= (lr / (1 - b1^t)) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)          -- nothing happens when skip is set
// Same formulas, in the same order, as torch's _single_tensor_adam.
//   nvsf_ema_update    shadow -= (1 - decay) * (shadow - param): torch_ema.ExponentialMovingAverage.update as the reference's
//                      Trainer runs it (trainer.py:112-114, 1420-1421), one pass per tensor; nvsf_adam_update takes the same
//                      shadow pointer to fold an every-step EMA into the optimiser pass (the parameter is in registers anyway).
#include "common.h"
#include <math.h>

namespace {
constexpr int kBlock = 256;
typedef float float4_t __attribute__((ext_vector_type(4)));

__global__ void k_adam_prepare(float* __restrict__ state, const float* __restrict__ found_inf, float beta1, float beta2) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const bool skip = found_inf && found_inf[0] != 0.0f;
    float step = state[0];
    if (!skip) step += 1.0f;
    state[0] = step;
    state[1] = (float)(1.0 - pow((double)beta1, (double)step));
    state[2] = (float)sqrt(1.0 - pow((double)beta2, (double)step));
    state[3] = skip ? 1.0f : 0.0f;
}

__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, float inv_scale_div, float step_size, float beta1, float beta2,
                                         float eps, float bc2_sqrt) {
    g = g / inv_scale_div;
    m = m + (1.0f - beta1) * (g - m);
    v = v * beta2 + (1.0f - beta2) * g * g;
    const float denom = sqrtf(v) / bc2_sqrt + eps;
    p = p - step_size * (m / denom);
}

__global__ __launch_bounds__(kBlock) void k_adam_update(float* __restrict__ param, const float* __restrict__ grad, float* __restrict__ exp_avg,
                                                        float* __restrict__ exp_avg_sq, unsigned long long n, float lr, float beta1, float beta2,
                                                        float eps, const float* __restrict__ state, const float* __restrict__ grad_scale,
                                                        float* __restrict__ ema, float ema_omd, _Float16* __restrict__ half_copy, int vec) {
    if (state[3] != 0.0f) return;  // overflow in this step's gradients: the step is skipped (GradScaler semantics)
    const float scale = grad_scale ? grad_scale[0] : 1.0f;
    const float step_size = lr / state[1], bc2_sqrt = state[2];
    const unsigned long long stride = (unsigned long long)gridDim.x * kBlock;
    unsigned long long i = (unsigned long long)blockIdx.x * kBlock + threadIdx.x;
    if (vec) {
        const unsigned long long n4 = n / 4;
        for (; i < n4; i += stride) {
            float4_t p = reinterpret_cast<float4_t*>(param)[i], m = reinterpret_cast<float4_t*>(exp_avg)[i], v = reinterpret_cast<float4_t*>(exp_avg_sq)[i];
            const float4_t g = reinterpret_cast<const float4_t*>(grad)[i];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float pk = p[k], mk = m[k], vk = v[k];
                adam_one(pk, g[k], mk, vk, scale, step_size, beta1, beta2, eps, bc2_sqrt);
                p[k] = pk; m[k] = mk; v[k] = vk;
            }
            reinterpret_cast<float4_t*>(param)[i] = p;
            if (half_copy) {  // the fp16 copy the forward kernels read (tables, MLP weights): written here instead of by a cast pass per step
                typedef _Float16 half4v __attribute__((ext_vector_type(4)));
                reinterpret_cast<half4v*>(half_copy)[i] = __builtin_convertvector(p, half4v);
            }
            if (ema) {
                float4_t e = reinterpret_cast<float4_t*>(ema)[i];
#pragma unroll
                for (int k = 0; k < 4; ++k) e[k] = e[k] - (e[k] - p[k]) * ema_omd;
                reinterpret_cast<float4_t*>(ema)[i] = e;
            }
            reinterpret_cast<float4_t*>(exp_avg)[i] = m;
            reinterpret_cast<float4_t*>(exp_avg_sq)[i] = v;
        }
        i = n4 * 4 + (unsigned long long)blockIdx.x * kBlock + threadIdx.x;  // tail
    }
    for (; i < n; i += stride) {
        adam_one(param[i], grad[i], exp_avg[i], exp_avg_sq[i], scale, step_size, beta1, beta2, eps, bc2_sqrt);
        if (half_copy) half_copy[i] = (_Float16)param[i];
        if (ema) ema[i] = ema[i] - (ema[i] - param[i]) * ema_omd;
    }
}

__global__ __launch_bounds__(kBlock) void k_ema_update(float* __restrict__ shadow, const float* __restrict__ param, unsigned long long n, float omd,
                                                       int vec) {
    const unsigned long long stride = (unsigned long long)gridDim.x * kBlock;
    unsigned long long i = (unsigned long long)blockIdx.x * kBlock + threadIdx.x;
    if (vec) {
        const unsigned long long n4 = n / 4;
        for (; i < n4; i += stride) {
            float4_t e = reinterpret_cast<float4_t*>(shadow)[i];
            const float4_t p = reinterpret_cast<const float4_t*>(param)[i];
#pragma unroll
            for (int k = 0; k < 4; ++k) e[k] = e[k] - (e[k] - p[k]) * omd;
            reinterpret_cast<float4_t*>(shadow)[i] = e;
        }
        i = n4 * 4 + (unsigned long long)blockIdx.x * kBlock + threadIdx.x;
    }
    for (; i < n; i += stride) shadow[i] = shadow[i] - (shadow[i] - param[i]) * omd;
}
}  // namespace

#define REQUIRE(cond) do { if (!(cond)) return NVSF_ERR_INVALID_ARG; } while (0)

NVSF_API int nvsf_adam_prepare(float* state4, const float* found_inf, float beta1, float beta2, hipStream_t stream) {
    REQUIRE(state4 && beta1 >= 0.0f && beta1 < 1.0f && beta2 >= 0.0f && beta2 < 1.0f);
    hipLaunchKernelGGL(k_adam_prepare, dim3(1), dim3(64), 0, stream, state4, found_inf, beta1, beta2);
    return nvsf_launch_status();
}

NVSF_API int nvsf_adam_update(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, uint64_t n, float lr, float beta1,
                              float beta2, float eps, const float* state4, const float* grad_scale, float* ema_shadow, float ema_one_minus_decay,
                              void* param_f16, hipStream_t stream) {
    if (n == 0) return NVSF_OK;
    REQUIRE(param && grad && exp_avg && exp_avg_sq && state4);
    const uintptr_t all = reinterpret_cast<uintptr_t>(param) | reinterpret_cast<uintptr_t>(grad) | reinterpret_cast<uintptr_t>(exp_avg) |
                          reinterpret_cast<uintptr_t>(exp_avg_sq) | reinterpret_cast<uintptr_t>(ema_shadow);
    const int vec = (all & 15u) == 0 && (reinterpret_cast<uintptr_t>(param_f16) & 7u) == 0;
    const unsigned long long work = vec ? (n + 3) / 4 : n;
    unsigned long long blocks = (work + kBlock - 1) / kBlock;
    if (blocks > 256ull * 16) blocks = 256ull * 16;  // grid-stride beyond 16 workgroups per CU
    hipLaunchKernelGGL(k_adam_update, dim3((unsigned)blocks), dim3(kBlock), 0, stream, param, grad, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps,
                       state4, grad_scale, ema_shadow, ema_one_minus_decay, reinterpret_cast<_Float16*>(param_f16), vec);
    return nvsf_launch_status();
}

NVSF_API int nvsf_ema_update(float* shadow, const float* param, uint64_t n, float one_minus_decay, hipStream_t stream) {
    if (n == 0) return NVSF_OK;
    REQUIRE(shadow && param && one_minus_decay >= 0.0f && one_minus_decay <= 1.0f);
    const int vec = ((reinterpret_cast<uintptr_t>(shadow) | reinterpret_cast<uintptr_t>(param)) & 15u) == 0;
    const unsigned long long work = vec ? (n + 3) / 4 : n;
    unsigned long long blocks = (work + kBlock - 1) / kBlock;
    if (blocks > 256ull * 16) blocks = 256ull * 16;
    hipLaunchKernelGGL(k_ema_update, dim3((unsigned)blocks), dim3(kBlock), 0, stream, shadow, param, n, one_minus_decay, vec);
    return nvsf_launch_status();
}
