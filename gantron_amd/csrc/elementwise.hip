// Bandwidth-bound kernels of the training step: BatchNorm(train)+activation+dropout forward and
// backward over channels-last rows, bias/activation/dropout, embedding gather/scatter, layout
// permutes, the output padding mask, the counter-based dropout RNG and the discriminator's
// window gather.  All are coalesced along the channel (fastest) dimension; per-channel reductions
// accumulate per-thread over a slab of rows, fold through LDS and finish with one fp32 atomic per
// (block, channel).
#include "common.h"

namespace {

constexpr int CR_TX = 64;     // channels per block
constexpr int CR_TY = 4;      // row lanes per block
constexpr int CR_ROWS = 256;  // rows per block

__device__ __forceinline__ float act_fwd(float z, int act) {
    return act == 1 ? fmaxf(z, 0.f) : (act == 2 ? tanhf(z) : z);
}

// fold two per-thread partials over the CR_TY row lanes.  ws == NULL: add them to out[c], out[C+c] with one fp32 atomic per
// (block, channel) -- the order of the row blocks, and with it the last bits of the sums, then depends on dispatch timing.
// ws != NULL (deterministic form): the block's partials go to ws[blockIdx.y][2][Cp] and the LAST row block of a channel
// group to arrive (ticket at ws + gridDim.y * 2 * Cp, zeroed by the launcher) adds all of them in row-block order -- the
// same bits whatever else runs on the device.  BatchNorm statistics are computed this way: activations within rounding of a
// ReLU threshold otherwise land on either side from one run (or one schedule) to the next, and one such element moves a
// channel's slice of every gradient below it (tools/enc_grad_probe.py).
// Hand-off form (MI355X_MICROARCH.md, Valid forms): only wave 0 stores; it drains its stores, one lane releases at agent
// scope and draws the ticket; the last arriver acquires at agent scope before its plain loads.
__device__ __forceinline__ void col_fold(float s0, float s1, int c, int C, float* out, float* ws) {
    __shared__ float red[2][CR_TY][CR_TX];
    __shared__ unsigned last;
    red[0][threadIdx.y][threadIdx.x] = s0;
    red[1][threadIdx.y][threadIdx.x] = s1;
    __syncthreads();
    if (threadIdx.y != 0) return;               // wave 0 finishes the job
    float a = 0.f, b = 0.f;
#pragma unroll
    for (int i = 0; i < CR_TY; ++i) { a += red[0][i][threadIdx.x]; b += red[1][i][threadIdx.x]; }
    if (ws == nullptr) {
        if (c < C) { atomicAdd(&out[c], a); atomicAdd(&out[C + c], b); }
        return;
    }
    const long Cp = (long)gridDim.x * CR_TX;
    float* mine = ws + (long)blockIdx.y * 2 * Cp;
    mine[c] = a;
    mine[Cp + c] = b;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        unsigned* ticket = reinterpret_cast<unsigned*>(ws + (long)gridDim.y * 2 * Cp) + blockIdx.x;
        last = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.y - 1 ? 1u : 0u;
        if (last) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
    __builtin_amdgcn_wave_barrier();            // one wave: lane 0's LDS store is visible to the others after this
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (!last || c >= C) return;
    a = b = 0.f;
    for (unsigned y = 0; y < gridDim.y; ++y) {
        a += __builtin_nontemporal_load(ws + (long)y * 2 * Cp + c);
        b += __builtin_nontemporal_load(ws + (long)y * 2 * Cp + Cp + c);
    }
    out[c] = a;
    out[C + c] = b;
}

// shift_first: sums of (x-K) and (x-K)^2 with K = x[0][c] (shifted-data variance); else sum x, sum x*y
__global__ __launch_bounds__(256) void col_stats_kernel(const float* x, const float* y, long rows, int C,
                                                        int shift_first, float* out, float* ws) {
    const int c = blockIdx.x * CR_TX + threadIdx.x;
    const long r0 = (long)blockIdx.y * CR_ROWS;
    const long r1 = min(rows, r0 + CR_ROWS);
    float s0 = 0.f, s1 = 0.f;
    if (c < C) {
        const float K = shift_first ? x[c] : 0.f;
        for (long r = r0 + threadIdx.y; r < r1; r += CR_TY) {
            const float xv = x[r * C + c] - K;
            const float yv = (y != nullptr) ? y[r * C + c] : xv;
            s0 += xv;
            s1 += xv * yv;
        }
    }
    col_fold(s0, s1, c, C, out, ws);
}

__global__ __launch_bounds__(256) void bn_fwd_kernel(const float* x, const float* stats, long rows, int C,
                                                     const float* gamma, const float* beta, float* rmean, float* rvar,
                                                     float momentum, float eps, int act, const uint8_t* keep,
                                                     float scale, float* y, float* mean_invstd, int training) {
    const long n = rows * C;
    const float inv_n = 1.0f / (float)rows;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (long)gridDim.x * blockDim.x) {
        const int c = (int)(idx % C);
        float mean, invstd;
        if (training) {
            const float K = x[c];
            const float s1 = stats[c] * inv_n;
            mean = K + s1;
            const float var = fmaxf(stats[C + c] * inv_n - s1 * s1, 0.f);
            invstd = rsqrtf(var + eps);
            if (idx < C) {   // one thread per channel publishes statistics
                mean_invstd[c] = mean;
                mean_invstd[C + c] = invstd;
                const float unbiased = var * ((float)rows / (float)max(rows - 1, 1L));
                rmean[c] = (1.f - momentum) * rmean[c] + momentum * mean;
                rvar[c] = (1.f - momentum) * rvar[c] + momentum * unbiased;
            }
        } else {
            mean = rmean[c];
            invstd = rsqrtf(rvar[c] + eps);
        }
        const float z = (x[idx] - mean) * invstd * gamma[c] + beta[c];
        float a = act_fwd(z, act);
        if (keep != nullptr) a *= keep[idx] ? scale : 0.f;
        y[idx] = a;
    }
}

// dz = dy * keep*scale * act'(z); column sums of dz and dz*xhat
__global__ __launch_bounds__(256) void bn_bwd1_kernel(const float* dy, const float* x, const float* mean_invstd,
                                                      const float* gamma, const float* beta, long rows, int C, int act,
                                                      const uint8_t* keep, float scale, float* dz, float* sums, float* ws) {
    const int c = blockIdx.x * CR_TX + threadIdx.x;
    const long r0 = (long)blockIdx.y * CR_ROWS;
    const long r1 = min(rows, r0 + CR_ROWS);
    float s0 = 0.f, s1 = 0.f;
    if (c < C) {
        const float mean = mean_invstd[c], invstd = mean_invstd[C + c], g = gamma[c], bt = beta[c];
        for (long r = r0 + threadIdx.y; r < r1; r += CR_TY) {
            const long i = r * C + c;
            const float xhat = (x[i] - mean) * invstd;
            const float z = xhat * g + bt;
            float d = dy[i];
            if (keep != nullptr) d *= keep[i] ? scale : 0.f;
            if (act == 1) d = z > 0.f ? d : 0.f;
            else if (act == 2) { const float t = tanhf(z); d *= (1.f - t * t); }
            dz[i] = d;
            s0 += d;
            s1 += d * xhat;
        }
    }
    col_fold(s0, s1, c, C, sums, ws);
}

__global__ __launch_bounds__(256) void bn_bwd2_kernel(const float* dz, const float* x, const float* mean_invstd,
                                                      const float* gamma, const float* sums, long rows, int C,
                                                      float* dx, float* dgamma, float* dbeta) {
    const long n = rows * C;
    const float inv_n = 1.0f / (float)rows;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (long)gridDim.x * blockDim.x) {
        const int c = (int)(idx % C);
        const float mean = mean_invstd[c], invstd = mean_invstd[C + c];
        const float xhat = (x[idx] - mean) * invstd;
        dx[idx] = gamma[c] * invstd * (dz[idx] - sums[c] * inv_n - xhat * sums[C + c] * inv_n);
        if (idx < C) {
            if (dbeta != nullptr) dbeta[c] += sums[c];
            if (dgamma != nullptr) dgamma[c] += sums[C + c];
        }
    }
}

__global__ __launch_bounds__(256) void bias_act_drop_fwd_kernel(const float* x, const float* bias, long n, int C,
                                                                int act, int order, const uint8_t* keep, float scale,
                                                                float* y) {
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (long)gridDim.x * blockDim.x) {
        float z = x[idx] + (bias != nullptr ? bias[idx % C] : 0.f);
        const float m = (keep != nullptr) ? (keep[idx] ? scale : 0.f) : 1.f;
        y[idx] = order == 0 ? act_fwd(z, act) * m : act_fwd(z * m, act);
    }
}

__global__ __launch_bounds__(256) void bias_act_drop_bwd_kernel(const float* dy, const float* y, long n, int act,
                                                                int order, const uint8_t* keep, float scale, float* dx) {
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (long)gridDim.x * blockDim.x) {
        const float m = (keep != nullptr) ? (keep[idx] ? scale : 0.f) : 1.f;
        float d = dy[idx];
        if (order == 0) {            // y = act(z) * m ; relu: act' = [y > 0]
            d *= m;
            if (act == 1) d = y[idx] > 0.f ? d : 0.f;
        } else {                     // y = act(z * m)
            if (act == 2) d *= (1.f - y[idx] * y[idx]);
            else if (act == 1) d = y[idx] > 0.f ? d : 0.f;
            d *= m;
        }
        dx[idx] = d;
    }
}

__global__ __launch_bounds__(256) void embedding_fwd_kernel(const int64_t* ids, const float* table, long n, int D,
                                                            float* out, long ld) {
    const long total = n * D;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const long i = idx / D;
        const int dd = (int)(idx - i * D);
        out[i * ld + dd] = table[ids[i] * D + dd];
    }
}

__global__ __launch_bounds__(256) void embedding_bwd_kernel(const int64_t* ids, const float* dout, long ld, long n,
                                                            int D, float* dtable) {
    const long total = n * D;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const long i = idx / D;
        const int dd = (int)(idx - i * D);
        atomicAdd(&dtable[ids[i] * D + dd], dout[i * ld + dd]);
    }
}

__global__ __launch_bounds__(256) void permute3_kernel(const float* in, long s0, long s1, long s2, int n0, int n1,
                                                       int n2, float* out, long o0, long o1, int accumulate,
                                                       float alpha) {
    const long total = (long)n0 * n1 * n2;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int k = (int)(idx % n2);
        const long ij = idx / n2;
        const int j = (int)(ij % n1);
        const long i = ij / n1;
        const float v = alpha * in[i * s0 + j * s1 + k * s2];
        const long o = i * o0 + j * o1 + k;
        out[o] = accumulate ? out[o] + v : v;
    }
}

// out[r][0..C) = idx[r] >= 0 ? src[idx[r]][0..C) : 0   (row gather; inverse maps give the scatter)
__global__ __launch_bounds__(256) void gather_rows_kernel(const float* src, long ld_src, const int* idx, long rows,
                                                          int C, float* out, long ld_out, int accumulate) {
    const long total = rows * C;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / C;
        const int c = (int)(i - r * C);
        const int sidx = idx[r];
        const float v = sidx >= 0 ? src[(long)sidx * ld_src + c] : 0.f;
        float* o = out + r * ld_out + c;
        *o = accumulate ? *o + v : v;
    }
}

__global__ __launch_bounds__(256) void axpby_kernel(const float* x, float a, const float* y, float b, float* out,
                                                    long n) {
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (long)gridDim.x * blockDim.x) {
        float v = 0.f;
        if (x != nullptr) v += a * x[idx];
        if (y != nullptr) v += b * y[idx];
        out[idx] = v;
    }
}

__global__ __launch_bounds__(256) void scale_dev_kernel(const float* x, const float* s, float host_scale, float* out,
                                                        long n, int accumulate) {
    const float f = (s != nullptr ? *s : 1.f) * host_scale;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (long)gridDim.x * blockDim.x)
        out[idx] = accumulate ? out[idx] + f * x[idx] : f * x[idx];
}

__global__ __launch_bounds__(256) void mask_outputs_kernel(float* mel, float* mel_post, float* gate,
                                                           const int64_t* out_len, int B, int n_mel, int To) {
    const long total = (long)B * n_mel * To;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int t = (int)(idx % To);
        const long bm = idx / To;
        const int m = (int)(bm % n_mel);
        const int b = (int)(bm / n_mel);
        if (t >= out_len[b]) {
            mel[idx] = 0.f;
            mel_post[idx] = 0.f;
            if (m == 0) gate[(long)b * To + t] = 1e3f;
        }
    }
}

__global__ __launch_bounds__(256) void keep_mask_kernel(uint64_t seed, uint64_t offset, float p, long n,
                                                        uint8_t* keep) {
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (long)gridDim.x * blockDim.x)
        keep[idx] = gt_u01(seed, offset + (uint64_t)idx) >= p ? 1 : 0;
}

__global__ __launch_bounds__(256) void uniform_kernel(uint64_t seed, uint64_t offset, long n, float* out) {
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (long)gridDim.x * blockDim.x)
        out[idx] = gt_u01(seed, offset + (uint64_t)idx);
}

// source frame of window-adjusted frame t' (model.py:523-524)
__device__ __forceinline__ int disc_src(int tp, int T, int Tm, int window) {
    return tp < Tm ? tp : T - window + (tp - Tm);
}

__global__ __launch_bounds__(256) void disc_gather_fwd_kernel(const float* mel, int B, int n_mel, int T, int window,
                                                              int W, float* r) {
    const int Cin = window * n_mel;
    const int Tm = T - T % window;
    const long total = (long)B * W * Cin;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int c = (int)(idx % Cin);
        const long bw = idx / Cin;
        const int w = (int)(bw % W);
        const int b = (int)(bw / W);
        const long f = (long)c * W + w;
        const int tp = (int)(f / n_mel), m = (int)(f % n_mel);
        r[idx] = mel[((long)b * n_mel + m) * T + disc_src(tp, T, Tm, window)];
    }
}

__global__ __launch_bounds__(256) void disc_gather_bwd_kernel(const float* dr, int B, int n_mel, int T, int window,
                                                              int W, float* dmel, int accumulate) {
    const int Cin = window * n_mel;
    const int Tm = T - T % window;
    const long total = (long)B * n_mel * T;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int t = (int)(idx % T);
        const long bm = idx / T;
        const int m = (int)(bm % n_mel);
        const int b = (int)(bm / n_mel);
        float g = 0.f;
        if (t < Tm) {
            const long f = (long)t * n_mel + m;
            g += dr[((long)b * W + (int)(f % W)) * Cin + (int)(f / W)];
        }
        if (Tm != T && t >= T - window) {
            const long f = (long)(Tm + t - (T - window)) * n_mel + m;
            g += dr[((long)b * W + (int)(f % W)) * Cin + (int)(f / W)];
        }
        dmel[idx] = accumulate ? dmel[idx] + g : g;
    }
}

__global__ __launch_bounds__(256) void adv_mean_fwd_kernel(const float* out, const int64_t* lengths, int B, int W,
                                                           int window, float* loss) {
    __shared__ float red[16];
    float acc = 0.f;
    for (int i = threadIdx.x; i < B * W; i += blockDim.x) {
        const int b = i / W, w = i - b * W;
        const int nb = (int)((lengths[b] + window - 1) / window);
        if (w < nb) acc += out[i] / (float)nb;
    }
    acc = block_sum(acc, red);
    if (threadIdx.x == 0) *loss = acc / (float)B;
}

__global__ __launch_bounds__(256) void adv_mean_bwd_kernel(const float* dloss, const int64_t* lengths, int B, int W,
                                                           int window, float scale, float* dout) {
    const float g = (dloss != nullptr ? *dloss : 1.f) * scale / (float)B;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < B * W; i += gridDim.x * blockDim.x) {
        const int b = i / W, w = i - b * W;
        const int nb = (int)((lengths[b] + window - 1) / window);
        dout[i] = w < nb ? g / (float)nb : 0.f;
    }
}

__global__ __launch_bounds__(256) void reflect_pad_kernel(const float* audio, int B, long n, int pad, float* out) {
    const long np = n + 2L * pad;
    const long total = (long)B * np;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const long b = idx / np;
        long i = idx - b * np - pad;
        if (i < 0) i = -i;
        if (i >= n) i = 2 * (n - 1) - i;
        out[idx] = audio[b * n + i];
    }
}

// Ragged batch: row b holds len[b] samples (ld apart); every row is reflect-padded around ITS OWN ends
// (so each item sees exactly what a single-item call would) and zero beyond len[b] + 2*pad.
__global__ __launch_bounds__(256) void reflect_pad_ragged_kernel(const float* audio, int B, long ld, const long* len,
                                                                 int pad, float* out) {
    const long np = ld + 2L * pad;
    const long total = (long)B * np;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const long b = idx / np;
        const long n = len[b];
        const long j = idx - b * np;
        float v = 0.f;
        if (j < n + 2L * pad) {
            long i = j - pad;
            if (i < 0) i = -i;
            if (i >= n) i = 2 * (n - 1) - i;
            v = audio[b * ld + i];
        }
        out[idx] = v;
    }
}

// log-mel rows (B*F, n_mel) -> mel_padded (B, n_mel, F) zero beyond frames[b], gate_padded (B, F) = 1 from
// frame frames[b]-1 on (TextMelCollate, data_utils.py:117-128)
__global__ __launch_bounds__(256) void collate_mel_kernel(const float* rows, const long* frames, int B, int n_mel, int F,
                                                          float* mel, float* gate) {
    const long total = (long)B * n_mel * F;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int f = (int)(idx % F);
        const int c = (int)((idx / F) % n_mel);
        const int b = (int)(idx / ((long)F * n_mel));
        const long nf = frames[b];
        mel[idx] = f < nf ? rows[((long)b * F + f) * n_mel + c] : 0.f;
        if (c == 0) gate[(long)b * F + f] = f >= nf - 1 ? 1.f : 0.f;
    }
}

__global__ __launch_bounds__(256) void stft_mag_kernel(const float* ft, long rows, int cutoff, float* mag) {
    const long total = rows * cutoff;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const long r = idx / cutoff;
        const int c = (int)(idx - r * cutoff);
        const float re = ft[r * 2 * cutoff + c], im = ft[r * 2 * cutoff + cutoff + c];
        mag[idx] = sqrtf(re * re + im * im);
    }
}

__global__ __launch_bounds__(256) void log_clamp_kernel(const float* x, long n, float clip, float* out) {
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (long)gridDim.x * blockDim.x)
        out[idx] = logf(fmaxf(x[idx], clip));
}

// ---- gradient penalty (train.py:32-69) -------------------------------------------------------------------------
// interpolated[b][c][t] = t < len_b ? alpha_b * real + (1 - alpha_b) * fake : 0      (train.py:36-49)
__global__ __launch_bounds__(256) void gp_interpolate_kernel(const float* real, const float* fake, const float* alpha,
                                                             const int64_t* lengths, int B, long CT, int T, float* out) {
    const long n = (long)B * CT;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (long)gridDim.x * blockDim.x) {
        const int b = (int)(idx / CT);
        const int t = (int)((idx - (long)b * CT) % T);
        const float a = alpha[b];
        out[idx] = t < lengths[b] ? a * real[idx] + (1.f - a) * fake[idx] : 0.f;
    }
}

// One workgroup per item: g_b <- masked gradient (train.py:60), n_b = sqrt(sum g_b^2 + 1e-12) (:67),
// gp += (n_b - 1)^2 / B (:70), and g_b is overwritten with d gp / d g_b = (2 / B) (n_b - 1) / n_b * g_b.
__global__ __launch_bounds__(256) void gp_norm_kernel(float* g, const int64_t* lengths, int B, long CT, int T, float* gp) {
    __shared__ float red[16];
    const int b = blockIdx.x;
    const int len = (int)lengths[b];
    float* gb = g + (long)b * CT;
    float s = 0.f;
    for (long i = threadIdx.x; i < CT; i += blockDim.x) {
        const float v = (int)(i % T) < len ? gb[i] : 0.f;
        s += v * v;
    }
    s = block_sum(s, red);
    const float nb = sqrtf(s + 1e-12f);
    const float coef = 2.0f / (float)B * (nb - 1.0f) / nb;
    if (threadIdx.x == 0) atomicAdd(gp, (nb - 1.0f) * (nb - 1.0f) / (float)B);
    for (long i = threadIdx.x; i < CT; i += blockDim.x) gb[i] = (int)(i % T) < len ? coef * gb[i] : 0.f;
}

// Second-order terms of a = tanh(m z) (DiscConv1d, layers.py:54-57) for the double backward of the penalty.  With
// t = da/dz = m (1 - a^2): the first backward computed dz = da * t; given p = d gp / d dz this returns
// u = d gp / d da = p * t and e = d gp / d z (through t) = p * da * (-2 m a t).
__global__ __launch_bounds__(256) void tanh_drop_bwd2_kernel(const float* p, const float* da, const float* y, long n,
                                                             const uint8_t* keep, float scale, float* u, float* e) {
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (long)gridDim.x * blockDim.x) {
        const float m = (keep != nullptr) ? (keep[idx] ? scale : 0.f) : 1.f;
        const float a = y[idx];
        const float t = m * (1.f - a * a);
        const float pv = p[idx];
        u[idx] = pv * t;
        e[idx] = pv * da[idx] * (-2.f * m * a * t);
    }
}

inline unsigned flat_blocks(long n) { return (unsigned)std::min<long>(std::max<long>(cdiv64(n, 256), 1), 8192); }

}  // namespace

#define ST ((hipStream_t)stream)

// floats of workspace the deterministic forms of gt_col_stats / gt_bn_act_drop_bwd1 need (partials + tickets)
extern "C" int64_t gt_col_ws_elems(int64_t rows, int32_t C) {
    const int64_t gx = cdiv(C, CR_TX), gy = cdiv64(rows, CR_ROWS);
    return gy * 2 * gx * CR_TX + gx;
}
// zero either the output (atomic form) or the tickets (deterministic form)
static int col_prepare(float* out2C, float* ws, int64_t rows, int32_t C, hipStream_t st, const char* who) {
    const int64_t gx = cdiv(C, CR_TX), gy = cdiv64(rows, CR_ROWS);
    hipError_t e = ws != nullptr ? hipMemsetAsync(ws + gy * 2 * gx * CR_TX, 0, sizeof(unsigned) * gx, st)
                                 : hipMemsetAsync(out2C, 0, sizeof(float) * 2 * C, st);
    GT_REQUIRE(e == hipSuccess, (int)e, "%s: memset failed", who);
    return 0;
}

extern "C" int gt_col_stats(const float* x, const float* y, int64_t rows, int32_t C, float* out2C, float* ws, void* stream) {
    GT_REQUIRE(x && out2C && rows > 0 && C > 0, GT_E_ARG, "gt_col_stats: bad argument");
    int rc = col_prepare(out2C, ws, rows, C, ST, "gt_col_stats");
    if (rc != 0) return rc;
    dim3 grid(cdiv(C, CR_TX), (unsigned)cdiv64(rows, CR_ROWS));
    // y == x selects the shifted-variance form used by BatchNorm forward
    const int shift_first = (y == x) ? 1 : 0;
    hipLaunchKernelGGL(col_stats_kernel, grid, dim3(CR_TX, CR_TY), 0, ST, x, shift_first ? nullptr : y, (long)rows, C,
                       shift_first, out2C, ws);
    GT_LAUNCH_CHECK();
    return 0;
}

extern "C" int gt_bn_act_drop_fwd(const float* x, const float* stats, int64_t rows, int32_t C, const float* gamma,
                                  const float* beta, float* running_mean, float* running_var, float momentum,
                                  float eps, int32_t act, const uint8_t* keep, float p_drop, float* y,
                                  float* mean_invstd, int32_t training, void* stream) {
    GT_REQUIRE(x && gamma && beta && running_mean && running_var && y && rows > 0 && C > 0, GT_E_ARG, "gt_bn_act_drop_fwd: bad argument");
    GT_REQUIRE(!training || (stats && mean_invstd), GT_E_ARG, "gt_bn_act_drop_fwd: training needs stats and mean_invstd");
    GT_REQUIRE(act >= 0 && act <= 2 && p_drop >= 0.f && p_drop < 1.f, GT_E_ARG, "gt_bn_act_drop_fwd: act/p_drop");
    hipLaunchKernelGGL(bn_fwd_kernel, dim3(flat_blocks(rows * C)), dim3(256), 0, ST, x, stats, (long)rows, C, gamma,
                       beta, running_mean, running_var, momentum, eps, act, keep, 1.0f / (1.0f - p_drop), y,
                       mean_invstd, training);
    GT_LAUNCH_CHECK();
    return 0;
}

extern "C" int gt_bn_act_drop_bwd1(const float* dy, const float* x, const float* mean_invstd, const float* gamma,
                                   const float* beta, int64_t rows, int32_t C, int32_t act, const uint8_t* keep,
                                   float p_drop, float* dz, float* sums2C, float* ws, void* stream) {
    GT_REQUIRE(dy && x && mean_invstd && gamma && beta && dz && sums2C && rows > 0 && C > 0, GT_E_ARG, "gt_bn_act_drop_bwd1: bad argument");
    int rc = col_prepare(sums2C, ws, rows, C, ST, "gt_bn_act_drop_bwd1");
    if (rc != 0) return rc;
    dim3 grid(cdiv(C, CR_TX), (unsigned)cdiv64(rows, CR_ROWS));
    hipLaunchKernelGGL(bn_bwd1_kernel, grid, dim3(CR_TX, CR_TY), 0, ST, dy, x, mean_invstd, gamma, beta, (long)rows, C,
                       act, keep, 1.0f / (1.0f - p_drop), dz, sums2C, ws);
    GT_LAUNCH_CHECK();
    return 0;
}

extern "C" int gt_bn_act_drop_bwd2(const float* dz, const float* x, const float* mean_invstd, const float* gamma,
                                   const float* sums2C, int64_t rows, int32_t C, float* dx, float* dgamma,
                                   float* dbeta, void* stream) {
    GT_REQUIRE(dz && x && mean_invstd && gamma && sums2C && dx && rows > 0 && C > 0, GT_E_ARG, "gt_bn_act_drop_bwd2: bad argument");
    hipLaunchKernelGGL(bn_bwd2_kernel, dim3(flat_blocks(rows * C)), dim3(256), 0, ST, dz, x, mean_invstd, gamma, sums2C,
                       (long)rows, C, dx, dgamma, dbeta);
    GT_LAUNCH_CHECK();
    return 0;
}

extern "C" int gt_bias_act_drop_fwd(const float* x, const float* bias, int64_t rows, int32_t C, int32_t act,
                                    int32_t order, const uint8_t* keep, float p_drop, float* y, void* stream) {
    GT_REQUIRE(x && y && rows > 0 && C > 0 && act >= 0 && act <= 2 && (order == 0 || order == 1), GT_E_ARG, "gt_bias_act_drop_fwd: bad argument");
    hipLaunchKernelGGL(bias_act_drop_fwd_kernel, dim3(flat_blocks(rows * C)), dim3(256), 0, ST, x, bias, (long)rows * C,
                       C, act, order, keep, 1.0f / (1.0f - p_drop), y);
    GT_LAUNCH_CHECK();
    return 0;
}

extern "C" int gt_bias_act_drop_bwd(const float* dy, const float* y, int64_t rows, int32_t C, int32_t act,
                                    int32_t order, const uint8_t* keep, float p_drop, float* dx, void* stream) {
    GT_REQUIRE(dy && y && dx && rows > 0 && C > 0, GT_E_ARG, "gt_bias_act_drop_bwd: bad argument");
    GT_REQUIRE(!(order == 0 && act == 2), GT_E_ARG, "gt_bias_act_drop_bwd: tanh-then-dropout is not used by the path");
    hipLaunchKernelGGL(bias_act_drop_bwd_kernel, dim3(flat_blocks(rows * C)), dim3(256), 0, ST, dy, y, (long)rows * C,
                       act, order, keep, 1.0f / (1.0f - p_drop), dx);
    GT_LAUNCH_CHECK();
    return 0;
}

extern "C" int gt_embedding_fwd(const int64_t* ids, const float* table, int64_t n, int32_t D, float* out,
                                int64_t ld_out, void* stream) {
    GT_REQUIRE(ids && table && out && n > 0 && D > 0 && ld_out >= D, GT_E_ARG, "gt_embedding_fwd: bad argument");
    hipLaunchKernelGGL(embedding_fwd_kernel, dim3(flat_blocks(n * D)), dim3(256), 0, ST, ids, table, (long)n, D, out, (long)ld_out);
    GT_LAUNCH_CHECK();
    return 0;
}

extern "C" int gt_embedding_bwd(const int64_t* ids, const float* dout, int64_t ld, int64_t n, int32_t D,
                                float* dtable, void* stream) {
    GT_REQUIRE(ids && dout && dtable && n > 0 && D > 0 && ld >= D, GT_E_ARG, "gt_embedding_bwd: bad argument");
    hipLaunchKernelGGL(embedding_bwd_kernel, dim3(flat_blocks(n * D)), dim3(256), 0, ST, ids, dout, (long)ld, (long)n, D, dtable);
    GT_LAUNCH_CHECK();
    return 0;
}

extern "C" int gt_gather_rows(const float* src, int64_t ld_src, const int32_t* idx, int64_t rows, int32_t C,
                              float* out, int64_t ld_out, int32_t accumulate, void* stream) {
    GT_REQUIRE(src && idx && out && rows > 0 && C > 0 && ld_src >= C && ld_out >= C, GT_E_ARG, "gt_gather_rows: bad argument");
    hipLaunchKernelGGL(gather_rows_kernel, dim3(flat_blocks(rows * C)), dim3(256), 0, ST, src, (long)ld_src, idx, (long)rows,
                       C, out, (long)ld_out, accumulate);
    GT_LAUNCH_CHECK();
    return 0;
}

extern "C" int gt_permute3(const float* in, int64_t s0, int64_t s1, int64_t s2, int32_t n0, int32_t n1, int32_t n2,
                           float* out, int64_t o0, int64_t o1, int32_t accumulate, float alpha, void* stream) {
    GT_REQUIRE(in && out && n0 > 0 && n1 > 0 && n2 > 0, GT_E_ARG, "gt_permute3: bad argument");
    if (o1 == 0) o1 = n2;
    if (o0 == 0) o0 = (int64_t)n1 * o1;
    hipLaunchKernelGGL(permute3_kernel, dim3(flat_blocks((long)n0 * n1 * n2)), dim3(256), 0, ST, in, (long)s0, (long)s1,
                       (long)s2, n0, n1, n2, out, (long)o0, (long)o1, accumulate, alpha);
    GT_LAUNCH_CHECK();
    return 0;
}

extern "C" int gt_axpby(const float* x, float a, const float* y, float b, float* out, int64_t n, void* stream) {
    GT_REQUIRE(out && n > 0, GT_E_ARG, "gt_axpby: bad argument");
    hipLaunchKernelGGL(axpby_kernel, dim3(flat_blocks(n)), dim3(256), 0, ST, x, a, y, b, out, (long)n);
    GT_LAUNCH_CHECK();
    return 0;
}

extern "C" int gt_scale_dev(const float* x, const float* s, float host_scale, float* out, int64_t n,
                            int32_t accumulate, void* stream) {
    GT_REQUIRE(x && out && n > 0, GT_E_ARG, "gt_scale_dev: bad argument");
    hipLaunchKernelGGL(scale_dev_kernel, dim3(flat_blocks(n)), dim3(256), 0, ST, x, s, host_scale, out, (long)n, accumulate);
    GT_LAUNCH_CHECK();
    return 0;
}

extern "C" int gt_mask_outputs(float* mel, float* mel_post, float* gate, const int64_t* out_len, int32_t B,
                               int32_t n_mel, int32_t To, void* stream) {
    GT_REQUIRE(mel && mel_post && gate && out_len && B > 0 && n_mel > 0 && To > 0, GT_E_ARG, "gt_mask_outputs: bad argument");
    hipLaunchKernelGGL(mask_outputs_kernel, dim3(flat_blocks((long)B * n_mel * To)), dim3(256), 0, ST, mel, mel_post, gate,
                       out_len, B, n_mel, To);
    GT_LAUNCH_CHECK();
    return 0;
}

extern "C" int gt_keep_mask(uint64_t seed, uint64_t offset, float p, int64_t n, uint8_t* keep, void* stream) {
    GT_REQUIRE(keep && n > 0 && p >= 0.f && p < 1.f, GT_E_ARG, "gt_keep_mask: bad argument");
    hipLaunchKernelGGL(keep_mask_kernel, dim3(flat_blocks(n)), dim3(256), 0, ST, seed, offset, p, (long)n, keep);
    GT_LAUNCH_CHECK();
    return 0;
}

extern "C" int gt_uniform(uint64_t seed, uint64_t offset, int64_t n, float* out, void* stream) {
    GT_REQUIRE(out && n > 0, GT_E_ARG, "gt_uniform: bad argument");
    hipLaunchKernelGGL(uniform_kernel, dim3(flat_blocks(n)), dim3(256), 0, ST, seed, offset, (long)n, out);
    GT_LAUNCH_CHECK();
    return 0;
}

static int disc_W(int T, int window) { return T / window + (T % window ? 1 : 0); }

extern "C" int gt_disc_gather_fwd(const float* mel, int32_t B, int32_t n_mel, int32_t T, int32_t window, float* r,
                                  void* stream) {
    GT_REQUIRE(mel && r && B > 0 && n_mel > 0 && window > 0 && T >= window, GT_E_ARG, "gt_disc_gather_fwd: bad argument (T=%d window=%d)", T, window);
    const int W = disc_W(T, window);
    hipLaunchKernelGGL(disc_gather_fwd_kernel, dim3(flat_blocks((long)B * W * window * n_mel)), dim3(256), 0, ST, mel, B,
                       n_mel, T, window, W, r);
    GT_LAUNCH_CHECK();
    return 0;
}

extern "C" int gt_disc_gather_bwd(const float* dr, int32_t B, int32_t n_mel, int32_t T, int32_t window, float* dmel,
                                  int32_t accumulate, void* stream) {
    GT_REQUIRE(dr && dmel && B > 0 && n_mel > 0 && window > 0 && T >= window, GT_E_ARG, "gt_disc_gather_bwd: bad argument");
    const int W = disc_W(T, window);
    hipLaunchKernelGGL(disc_gather_bwd_kernel, dim3(flat_blocks((long)B * n_mel * T)), dim3(256), 0, ST, dr, B, n_mel, T,
                       window, W, dmel, accumulate);
    GT_LAUNCH_CHECK();
    return 0;
}

extern "C" int gt_adv_mean_fwd(const float* out, const int64_t* lengths, int32_t B, int32_t W, int32_t window,
                               float* loss, void* stream) {
    GT_REQUIRE(out && lengths && loss && B > 0 && W > 0 && window > 0, GT_E_ARG, "gt_adv_mean_fwd: bad argument");
    hipLaunchKernelGGL(adv_mean_fwd_kernel, dim3(1), dim3(256), 0, ST, out, lengths, B, W, window, loss);
    GT_LAUNCH_CHECK();
    return 0;
}

extern "C" int gt_adv_mean_bwd(const float* dloss, const int64_t* lengths, int32_t B, int32_t W, int32_t window,
                               float scale, float* dout, void* stream) {
    GT_REQUIRE(lengths && dout && B > 0 && W > 0 && window > 0, GT_E_ARG, "gt_adv_mean_bwd: bad argument");
    hipLaunchKernelGGL(adv_mean_bwd_kernel, dim3(flat_blocks((long)B * W)), dim3(256), 0, ST, dloss, lengths, B, W,
                       window, scale, dout);
    GT_LAUNCH_CHECK();
    return 0;
}

extern "C" int gt_reflect_pad(const float* audio, int32_t B, int64_t n, int32_t pad, float* out, void* stream) {
    GT_REQUIRE(audio && out && B > 0 && n > pad && pad >= 0, GT_E_ARG, "gt_reflect_pad: need n > pad");
    hipLaunchKernelGGL(reflect_pad_kernel, dim3(flat_blocks((long)B * (n + 2L * pad))), dim3(256), 0, ST, audio, B, (long)n, pad, out);
    GT_LAUNCH_CHECK();
    return 0;
}

extern "C" int gt_reflect_pad_ragged(const float* audio, int32_t B, int64_t ld, const int64_t* lengths, int32_t pad,
                                     float* out, void* stream) {
    GT_REQUIRE(audio && out && lengths && B > 0 && ld > pad && pad >= 0, GT_E_ARG, "gt_reflect_pad_ragged: bad argument");
    hipLaunchKernelGGL(reflect_pad_ragged_kernel, dim3(flat_blocks((long)B * (ld + 2L * pad))), dim3(256), 0, ST, audio, B,
                       (long)ld, (const long*)lengths, pad, out);
    GT_LAUNCH_CHECK();
    return 0;
}

extern "C" int gt_collate_mel(const float* rows, const int64_t* frames, int32_t B, int32_t n_mel, int32_t F, float* mel,
                              float* gate, void* stream) {
    GT_REQUIRE(rows && frames && mel && gate && B > 0 && n_mel > 0 && F > 0, GT_E_ARG, "gt_collate_mel: bad argument");
    hipLaunchKernelGGL(collate_mel_kernel, dim3(flat_blocks((long)B * n_mel * F)), dim3(256), 0, ST, rows, (const long*)frames, B,
                       n_mel, F, mel, gate);
    GT_LAUNCH_CHECK();
    return 0;
}

extern "C" int gt_stft_magnitude(const float* ft, int64_t rows, int32_t cutoff, float* mag, void* stream) {
    GT_REQUIRE(ft && mag && rows > 0 && cutoff > 0, GT_E_ARG, "gt_stft_magnitude: bad argument");
    hipLaunchKernelGGL(stft_mag_kernel, dim3(flat_blocks(rows * cutoff)), dim3(256), 0, ST, ft, (long)rows, cutoff, mag);
    GT_LAUNCH_CHECK();
    return 0;
}

extern "C" int gt_log_clamp(const float* x, int64_t n, float clip, float* out, void* stream) {
    GT_REQUIRE(x && out && n > 0, GT_E_ARG, "gt_log_clamp: bad argument");
    hipLaunchKernelGGL(log_clamp_kernel, dim3(flat_blocks(n)), dim3(256), 0, ST, x, (long)n, clip, out);
    GT_LAUNCH_CHECK();
    return 0;
}

extern "C" int gt_gp_interpolate(const float* real, const float* fake, const float* alpha, const int64_t* lengths,
                                 int32_t B, int32_t n_mel, int32_t T, float* out, void* stream) {
    GT_REQUIRE(real && fake && alpha && lengths && out && B > 0 && n_mel > 0 && T > 0, GT_E_ARG, "gt_gp_interpolate: bad argument");
    const long CT = (long)n_mel * T;
    hipLaunchKernelGGL(gp_interpolate_kernel, dim3(flat_blocks((long)B * CT)), dim3(256), 0, ST, real, fake, alpha, lengths,
                       B, CT, T, out);
    GT_LAUNCH_CHECK();
    return 0;
}

extern "C" int gt_gp_norm(float* g, const int64_t* lengths, int32_t B, int32_t n_mel, int32_t T, float* gp, void* stream) {
    GT_REQUIRE(g && lengths && gp && B > 0 && n_mel > 0 && T > 0, GT_E_ARG, "gt_gp_norm: bad argument");
    hipLaunchKernelGGL(gp_norm_kernel, dim3(B), dim3(256), 0, ST, g, lengths, B, (long)n_mel * T, T, gp);
    GT_LAUNCH_CHECK();
    return 0;
}

extern "C" int gt_tanh_drop_bwd2(const float* p, const float* da, const float* y, int64_t rows, int32_t C,
                                 const uint8_t* keep, float p_drop, float* u, float* e, void* stream) {
    GT_REQUIRE(p && da && y && u && e && rows > 0 && C > 0, GT_E_ARG, "gt_tanh_drop_bwd2: bad argument");
    hipLaunchKernelGGL(tanh_drop_bwd2_kernel, dim3(flat_blocks(rows * C)), dim3(256), 0, ST, p, da, y, (long)rows * C, keep,
                       1.0f / (1.0f - p_drop), u, e);
    GT_LAUNCH_CHECK();
    return 0;
}
