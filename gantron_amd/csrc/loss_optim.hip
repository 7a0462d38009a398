// Losses of the generator step and the fused optimiser update.
//   gt_taco_loss        : Tacotron2Loss (loss_function.py:10-22) value + gradients in one pass
//   gt_guided_attn_loss : attention_loss (loss_function.py:25-52) -- the reference runs a Python
//                         double loop of tiny ops per character; here one launch covers the batch
//   gt_sumsq / gt_adam_step : clip_grad_norm_ + Adam(weight_decay) (train.py:242-243,404-406)
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void taco_loss_kernel(const float* mel, const float* post, const float* gate,
                                                        const float* mel_t, const float* gate_t, long n_mel, long n_gate,
                                                        float* sums, float* dmel, float* dpost, float* dgate) {
    __shared__ float red[16];
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    const float inv_m = 1.0f / (float)n_mel, inv_g = 1.0f / (float)n_gate;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n_mel; i += (long)gridDim.x * blockDim.x) {
        const float t = mel_t[i];
        const float d0 = mel[i] - t, d1 = post[i] - t;
        s0 += d0 * d0;
        s1 += d1 * d1;
        dmel[i] = 2.f * d0 * inv_m;
        dpost[i] = 2.f * d1 * inv_m;
    }
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n_gate; i += (long)gridDim.x * blockDim.x) {
        const float x = gate[i], y = gate_t[i];
        // BCEWithLogits: max(x,0) - x*y + log(1 + exp(-|x|))
        s2 += fmaxf(x, 0.f) - x * y + log1pf(__expf(-fabsf(x)));
        dgate[i] = (sigmoidf_(x) - y) * inv_g;
    }
    s0 = block_sum(s0, red);
    s1 = block_sum(s1, red);
    s2 = block_sum(s2, red);
    if (threadIdx.x == 0) {
        atomicAdd(&sums[0], s0 * inv_m);
        atomicAdd(&sums[1], s1 * inv_m);
        atomicAdd(&sums[2], s2 * inv_g);
    }
}

// grid (chunks of To, B); align (B, To, Ti) row-major.
__global__ __launch_bounds__(256) void guided_attn_kernel(const float* align, const int64_t* in_len,
                                                          const int64_t* out_len, int B, int To, int Ti, float* loss,
                                                          float* dalign) {
    __shared__ float red[16];
    const int b = blockIdx.y;
    const int ni = (int)in_len[b], no = (int)out_len[b];
    const float w = 1.0f / ((float)ni * (float)no * (float)B);
    const int rows_per = cdiv(To, (int)gridDim.x);
    const int t0 = blockIdx.x * rows_per, t1 = min(To, t0 + rows_per);
    float acc = 0.f;
    for (int i = threadIdx.x; i < (t1 - t0) * Ti; i += blockDim.x) {
        const int t = t0 + i / Ti, n = i % Ti;
        const long idx = ((long)b * To + t) * Ti + n;
        float g = 0.f;
        if (t < no && n < ni) {
            const int c = (int)(((long)n * (no - 1)) / max(ni - 1, 1));
            const float dt = (float)t - (float)c;
            const float y = __expf(-(dt * dt) / 1.5f);
            const float x = align[idx];
            const float lx = fmaxf(__logf(x), -100.f), l1x = fmaxf(log1pf(-x), -100.f);
            acc += -(y * lx + (1.f - y) * l1x);
            g = (x - y) / fmaxf((1.f - x) * x, 1e-12f) * w;
        }
        dalign[idx] = g;
    }
    acc = block_sum(acc, red);
    if (threadIdx.x == 0) atomicAdd(loss, acc * w);
}

__global__ __launch_bounds__(256) void sumsq_kernel(const float* x, long n, float* out) {
    __shared__ float red[16];
    float s = 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) s += x[i] * x[i];
    s = block_sum(s, red);
    if (threadIdx.x == 0) atomicAdd(out, s);
}

__global__ __launch_bounds__(256) void adam_kernel(float* p, const float* g, float* m, float* v, long n, float lr,
                                                   float b1, float b2, float eps, float wd, float bc1, float bc2_sqrt,
                                                   const float* sumsq, float max_norm, const int* skip_flag) {
    // skip_flag (or NULL): a word that kernels with a bounded in-launch wait raise when a wait gives up (the gradients of that
    // step are then undefined): the step is not applied -- parameters and moments stay as they were -- without the host
    // having to look (it finds the flag at its next synchronisation point and stops)
    if (skip_flag != nullptr && __hip_atomic_load(skip_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return;
    float coef = 1.f;
    if (sumsq != nullptr) coef = fminf(max_norm / (sqrtf(*sumsq) + 1e-6f), 1.f);
    const float step = lr / bc1;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float pv = p[i];
        const float gr = g[i] * coef + wd * pv;
        const float mi = b1 * m[i] + (1.f - b1) * gr;
        const float vi = b2 * v[i] + (1.f - b2) * gr * gr;
        m[i] = mi;
        v[i] = vi;
        p[i] = pv - step * mi / (sqrtf(vi) / bc2_sqrt + eps);
    }
}

}  // namespace

#define ST ((hipStream_t)stream)

extern "C" int gt_taco_loss(const float* mel, const float* mel_post, const float* gate, const float* mel_t,
                            const float* gate_t, int64_t n_mel_elems, int64_t n_gate, float* sums3, float* dmel,
                            float* dmel_post, float* dgate, void* stream) {
    GT_REQUIRE(mel && mel_post && gate && mel_t && gate_t && sums3 && dmel && dmel_post && dgate && n_mel_elems > 0 && n_gate > 0,
               GT_E_ARG, "gt_taco_loss: bad argument");
    hipError_t e = hipMemsetAsync(sums3, 0, 3 * sizeof(float), ST);
    GT_REQUIRE(e == hipSuccess, (int)e, "gt_taco_loss: memset failed");
    const unsigned blocks = (unsigned)std::min<long>(cdiv64(n_mel_elems, 1024), 2048);
    hipLaunchKernelGGL(taco_loss_kernel, dim3(std::max(blocks, 1u)), dim3(256), 0, ST, mel, mel_post, gate, mel_t, gate_t,
                       (long)n_mel_elems, (long)n_gate, sums3, dmel, dmel_post, dgate);
    GT_LAUNCH_CHECK();
    return 0;
}

extern "C" int gt_guided_attn_loss(const float* align, const int64_t* in_len, const int64_t* out_len, int32_t B,
                                   int32_t To, int32_t Ti, float* loss, float* dalign, void* stream) {
    GT_REQUIRE(align && in_len && out_len && loss && dalign && B > 0 && To > 0 && Ti > 0, GT_E_ARG, "gt_guided_attn_loss: bad argument");
    hipError_t e = hipMemsetAsync(loss, 0, sizeof(float), ST);
    GT_REQUIRE(e == hipSuccess, (int)e, "gt_guided_attn_loss: memset failed");
    dim3 grid(std::min(To, 32), B);
    hipLaunchKernelGGL(guided_attn_kernel, grid, dim3(256), 0, ST, align, in_len, out_len, B, To, Ti, loss, dalign);
    GT_LAUNCH_CHECK();
    return 0;
}

extern "C" int gt_sumsq(const float* x, int64_t n, float* out, void* stream) {
    GT_REQUIRE(x && out && n > 0, GT_E_ARG, "gt_sumsq: bad argument");
    const unsigned blocks = (unsigned)std::min<long>(std::max<long>(cdiv64(n, 2048), 1), 1024);
    hipLaunchKernelGGL(sumsq_kernel, dim3(blocks), dim3(256), 0, ST, x, (long)n, out);
    GT_LAUNCH_CHECK();
    return 0;
}

extern "C" int gt_adam_step(float* param, const float* grad, float* m, float* v, int64_t n, float lr, float beta1,
                            float beta2, float eps, float weight_decay, int32_t step, const float* sumsq,
                            float max_norm, void* stream) {
    return gt_adam_step_guarded(param, grad, m, v, n, lr, beta1, beta2, eps, weight_decay, step, sumsq, max_norm, nullptr, stream);
}

extern "C" int gt_adam_step_guarded(float* param, const float* grad, float* m, float* v, int64_t n, float lr, float beta1,
                                    float beta2, float eps, float weight_decay, int32_t step, const float* sumsq,
                                    float max_norm, const int32_t* skip_flag, void* stream) {
    GT_REQUIRE(param && grad && m && v && n > 0 && step >= 1, GT_E_ARG, "gt_adam_step: bad argument");
    const float bc1 = 1.f - powf(beta1, (float)step);
    const float bc2s = sqrtf(1.f - powf(beta2, (float)step));
    const unsigned blocks = (unsigned)std::min<long>(std::max<long>(cdiv64(n, 1024), 1), 4096);
    hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(256), 0, ST, param, grad, m, v, (long)n, lr, beta1, beta2, eps,
                       weight_decay, bc1, bc2s, sumsq, max_norm, skip_flag);
    GT_LAUNCH_CHECK();
    return 0;
}
