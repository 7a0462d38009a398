// Body of the one-launch forward attention step (gt_attn_step2_fwd), included TEXTUALLY into attn_step2_fwd_mfma and
// attn_step2_tail_fwd_mfma (attention_mfma.hip): `a` is the kernel's AttnStep2Fwd argument, `S` its SharedM.  It is not a
// function on purpose -- see the note at the kernels.  A workgroup leaves the step through AT2_DONE.
    const GtAttnDims& d = a.d;
    // workgroup id -> (row, chunk): groups of 8 rows; inside a group id = chunk * 8 + (row % 8): a row's chunks share an XCD
    const int L = blockIdx.x;
    const int grp = L / (8 * a.NC), rem = L - grp * 8 * a.NC;
    const int chunk = rem >> 3, b = grp * 8 + (rem & 7);
    if (b >= d.B) AT2_DONE;
    const int t0 = chunk * CT;
    const int Tv = a.lengths != nullptr ? min(a.lengths[b], d.Ti) : d.Ti;      // valid positions of this row
    if (t0 >= Tv) AT2_DONE;                                                    // chunk entirely in the padding: not a member
    const int NCv = (Tv + CT - 1) / CT;
    long long tstamp[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    (void)tstamp;
    AT_STAMP(0);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, kq = lane >> 4;
    const int E4 = d.E >> 2;
    const int c4 = min(tid, E4 - 1);
    // ---- 1. energies of positions [t0, t0 + 32)  (as attn_energies_fwd_mfma).  Every input of the phase is REQUESTED before the
    // first one is waited for (one memory round trip instead of two: the query slabs of the LSTM launch that has just ended
    // come from farthest away and are requested last, loads return in order)
    constexpr int NWP = (WPACK_F4 + 255) / 256;
    float4 wp4[NWP];
    float v_in = 0.f, win_in = 0.f;
    {       // (the packed [Wc | Wd] image is required: gt_attn_step2_fwd checks it)
        const float4* src = reinterpret_cast<const float4*>(a.wpack);
#pragma unroll
        for (int i2 = 0; i2 < NWP; ++i2) wp4[i2] = src[min(tid + 256 * i2, WPACK_F4 - 1)];
        v_in = a.v[min(tid, A - 1)];
        const int tw2 = min(tid, 2 * WL - 1);
        const int c = tw2 / WL, p = tw2 - c * WL;
        const int t = t0 + p - (KC - 1) / 2;
        const float wv = (c == 0 ? a.w_prev : a.w_cum)[(long)b * d.Ti + min(max(t, 0), d.Ti - 1)];
        win_in = (p < CT + KC - 1 && t >= 0 && t < d.Ti) ? wv : 0.f;
    }
    float4 pm4[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int f = tid + 256 * i, tl = f >> 5, cc = f & 31;
        pm4[i] = *reinterpret_cast<const float4*>(a.pm + ((long)b * d.Ti + min(t0 + tl, d.Ti - 1)) * A + 4 * cc);
    }
    const bool qvec = ((a.qpart.ld | a.qpart.slab_stride) & 3) == 0 && (reinterpret_cast<uintptr_t>(a.qpart.p) & 15) == 0;
    float qv = 0.f;
    if (qvec) {
        const int sg = tid >> 5, cc = tid & 31;
        const float* qp = a.qpart.p + (long)b * a.qpart.ld + 4 * cc;
        float4 qa = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int base = 0; base < a.qpart.n; base += 64) {
            float4 t[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) t[j] = *reinterpret_cast<const float4*>(qp + (long)min(base + sg + 8 * j, a.qpart.n - 1) * a.qpart.slab_stride);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float w = base + sg + 8 * j < a.qpart.n ? 1.f : 0.f;
                qa.x += w * t[j].x; qa.y += w * t[j].y; qa.z += w * t[j].z; qa.w += w * t[j].w;
            }
        }
        *reinterpret_cast<float4*>(&S.g[sg * A + 4 * cc]) = qa;
    } else {
        const float* qp = a.qpart.p + (long)b * a.qpart.ld + (tid & (A - 1));
#pragma unroll 8
        for (int s2 = 0; s2 < a.qpart.n; ++s2) qv += qp[(long)s2 * a.qpart.slab_stride];
    }
    // (no stamp here: one between the slab sums and the staging makes the compiler move a register array into LDS, 80 KB
    // instead of 64, and the kernel then runs one workgroup per CU)
    {
        float4* dst = reinterpret_cast<float4*>(&S.wc[0]);
#pragma unroll
        for (int i2 = 0; i2 < NWP; ++i2) dst[min(tid + 256 * i2, WPACK_F4 - 1)] = wp4[i2];
        if (tid < A) S.v[tid] = v_in;
        if (tid < 2 * WL) S.win[tid / WL][tid % WL] = win_in;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int f = tid + 256 * i, tl = f >> 5, cc = f & 31;
        *reinterpret_cast<float4*>(&S.ds[tl * LDS_ + 4 * cc]) = pm4[i];
    }
    if (qvec) {
        __syncthreads();
        if (tid < A) {
            qv = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) qv += S.g[j * A + tid];
        }
    }
    if (tid < A) {
        S.q[tid] = qv;
        if (chunk == 0 && a.q_out != nullptr) a.q_out[(long)b * A + tid] = qv;
    }
    // ---- 0. the memory tile: thread c4 owns float4 column c4 of the chunk's 32 rows (E <= 1024: host check).  Requested
    // HERE, behind the energies phase's own loads (vector loads return in order: issued first, the 128 KB tile would hold
    // up the query slabs and the processed-memory tile the phase starts with), and consumed after the phase.
    float4 mem4[CT];
    {
        const float* mp = a.memory + ((long)b * d.Ti) * a.ld_mem + 4 * c4;
        // (Measured and not adopted: non-temporal loads here and non-temporal stores of u_out, so that the tile would not displace
        // the LSTM weights in L2 -- this kernel 19.9 us instead of 18.6 in the loop, the LSTM launch unchanged.  Knock-outs with the
        // phase stamps: without these 32 requests per thread the inputs are staged after 2.9 us instead of 4.7 and the last arriver is
        // done 2.1 us earlier; without the query slabs 0.4 us earlier, without the packed weights 0.2.  Half of the requests behind
        // the staging barrier: staged after 3.5 us, convolution done 5.2 instead of 5.4, the end unchanged -- the tile's 29 MB per
        // step cost their 2 us wherever they are asked for.)
#pragma unroll
        for (int l = 0; l < CT; ++l)
            mem4[l] = (a.dbg & 1) ? make_float4(0.f, 0.f, 0.f, 0.f) : *reinterpret_cast<const float4*>(mp + (long)min(t0 + l, Tv - 1) * a.ld_mem);
    }
    __syncthreads();
    AT_STAMP(2);                     // inputs staged, memory tile requested
    conv_tile(S, wave, lane);
    __syncthreads();
    AT_STAMP(3);                     // location convolution done
    f32x4 acc[2][2];
    dense_tiles(S, wave, lane, acc);
    float ep[2][4];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float sum = 0.f;
#pragma unroll
            for (int nl = 0; nl < 2; ++nl) {
                const int dd = 32 * wave + 16 * nl + r;
                const float u = tanhf_(acc[mt][nl][e] + S.q[dd] + S.ds[(16 * mt + 4 * kq + e) * LDS_ + dd]);
                sum += S.v[dd] * u;
                if (a.u_out != nullptr && t0 + 16 * mt + 4 * kq + e < d.Ti)
                    a.u_out[((long)b * d.Ti + t0 + 16 * mt + 4 * kq + e) * A + dd] = u;
            }
            sum += __shfl_xor(sum, 1, 64);
            sum += __shfl_xor(sum, 2, 64);
            sum += __shfl_xor(sum, 4, 64);
            sum += __shfl_xor(sum, 8, 64);
            ep[mt][e] = sum;
        }
    if (r == 0) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int e = 0; e < 4; ++e) S.epart[wave][16 * mt + 4 * kq + e] = ep[mt][e];
    }
    __syncthreads();
    if (a.dbg & 2) AT2_DONE;
    AT_STAMP(4);                     // energies done
    // ---- 2. softmax numerators relative to the chunk's own maximum (wave 0), then the partial context
    float* pl = S.de;                                   // CT numerators
    if (wave == 0) {
        const bool valid = lane < CT && t0 + lane < Tv;
        const float e = valid ? (S.epart[0][lane & 31] + S.epart[1][lane & 31]) + (S.epart[2][lane & 31] + S.epart[3][lane & 31]) : -INFINITY;
        const float m = wave_max(e);                    // finite: position t0 is valid
        const float pv = valid ? __expf(e - m) : 0.f;
        const float sm = wave_sum(pv);
        if (lane < CT) pl[lane] = pv;
        if (lane == 0) { S.red[0] = m; S.red[1] = sm; }
    }
    __syncthreads();
    float4 part = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int l = 0; l < CT; ++l) {
        const float pv = pl[l];
        part.x += pv * mem4[l].x; part.y += pv * mem4[l].y; part.z += pv * mem4[l].z; part.w += pv * mem4[l].w;
    }
    AT_STAMP(5);                     // partial context done (the memory tile has arrived)
    // ---- 3. publish (write-through), drain, ticket
    float* xrow = a.xbuf + ((long)b * a.NC + chunk) * a.xs;      // [E partial context | 32 numerators | m | s | pad]
    {
        const __amdgpu_buffer_rsrc_t wr = at_rsrc(xrow);
        if (tid < E4) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(au32x4, part), wr, tid * 16, 0, 16);
        if (wave == 3 && lane < 9) {                    // 8 x 16 B of numerators + one of (m, s)
            const float4 v4 = lane < 8 ? *reinterpret_cast<const float4*>(&pl[4 * lane]) : make_float4(S.red[0], S.red[1], 0.f, 0.f);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(au32x4, v4), wr, (d.E + 4 * lane) * 4, 0, 16);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // EVERY storing wave drains (its u_out stores too)
    __syncthreads();
    if (tid == 0) {
        const unsigned ticket = __hip_atomic_fetch_add(a.row_cnt + b, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        S.red[2] = ticket == (unsigned)NCv * a.step1 - 1u ? 1.f : 0.f;
    }
    __syncthreads();
    AT_STAMP(6);                     // published, ticket drawn
    if (S.red[2] == 0.f || (a.dbg & 4)) AT2_DONE;                    // not the row's last arriver
    // ---- 4. last arriver: everything the other members published is requested at once (one round trip): their
    // statistics, their partial contexts, their numerators
    float* stat = S.g;                                             // [NCv][2] (m_c, s_c)
    const float* xb = a.xbuf + (long)b * a.NC * a.xs;
    float4 ms = make_float4(0.f, 0.f, 0.f, 0.f);
    if (tid < NCv) {
        const __amdgpu_buffer_rsrc_t rd = at_rsrc(xb + (long)tid * a.xs);
        ms = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rd, (d.E + CT) * 4, 0, 16));
    }
    // (every member INCLUDING this one, whose partial context is in registers: the sum below runs in chunk order whoever
    // arrived last, so the step's result does not depend on the arrival order -- round 4)
    float4 others[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const int cc = min(c, NCv - 1);
        const __amdgpu_buffer_rsrc_t rd = at_rsrc(xb + (long)cc * a.xs);
        others[c] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rd, c4 * 16, 0, 16));
    }
    const int tw = min(tid, Tv - 1), cw = tw >> 5;                  // position tid of the row (Ti <= 256)
    float pw = __hip_atomic_load(xb + (long)cw * a.xs + d.E + (tw & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (cw == chunk) pw = pl[tw & 31];
    if (tid < NCv) {
        stat[2 * tid] = tid == chunk ? S.red[0] : ms.x;
        stat[2 * tid + 1] = tid == chunk ? S.red[1] : ms.y;
    }
    __syncthreads();
    float M = -INFINITY;
    for (int c = 0; c < NCv; ++c) M = fmaxf(M, stat[2 * c]);
    float Stot = 0.f;
    for (int c = 0; c < NCv; ++c) Stot += __expf(stat[2 * c] - M) * stat[2 * c + 1];
    const float inv = 1.0f / Stot;
    {
        float4 ctx = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const float so = c < NCv ? __expf(stat[2 * min(c, NCv - 1)] - M) : 0.f;
            const float4 t = c == chunk ? part : others[c];       // (its own bytes as computed, not as read back)
            ctx.x += so * t.x; ctx.y += so * t.y; ctx.z += so * t.z; ctx.w += so * t.w;
        }
        ctx.x *= inv; ctx.y *= inv; ctx.z *= inv; ctx.w *= inv;
        if (tid < E4) {
            const int col = 4 * tid;
            if (a.c0 != nullptr) *reinterpret_cast<float4*>(a.c0 + (long)b * a.ld0 + col) = ctx;
            if (a.c1 != nullptr) *reinterpret_cast<float4*>(a.c1 + (long)b * a.ld1 + col) = ctx;
            if (a.c2 != nullptr) *reinterpret_cast<float4*>(a.c2 + (long)b * a.ld2 + col) = ctx;
        }
    }
    if (tid < d.Ti) {
        const float wt = tid < Tv ? pw * __expf(stat[2 * cw] - M) * inv : 0.f;
        a.w_out[(long)b * d.Ti + tid] = wt;
        if (a.cum_out != nullptr) a.cum_out[(long)b * d.Ti + tid] = a.cum_in[(long)b * d.Ti + tid] + wt;
    }
    AT_STAMP(7);                     // last arriver: context, weights, cumulative weights written
