// tools/exp_cohort_fused4_kernel.h — EXPERIMENT, part of no build.  Round 3's seventh attempt at the AS-norm statistics kernel
// (neuralplda_amd/csrc/nplda_cohort_fused.hip, cohort_fused2_kernel): the two waves of a SIMD SPECIALISED — a matrix wave that
// issues nothing but MFMAs and a statistics wave that runs the previous tile's epilogue from a copy handed over through LDS.
// Drop it in behind cohort_fused2_kernel (it uses that file's FusedArgs + `int abl; unsigned long long* stamps;`, f32x4,
// kSubSlack; blocks of 192 rows: fa.ny = ceil(R / 192); launch with dim3(512)).  D <= 160 (NB <= 10) passes
// tests/test_cohort_fused_gpu.py + tests/test_asnorm_gpu.py; NB = 11 / 12 need more than the 128 AGPRs a wave gets at two
// waves per SIMD for the rows' operands and are WRONG as written.  It is NOT faster: 732 - 780 us against 682 us
// (cfg3, same box).
//
// What was measured (cycle stamps of block 0, s_memtime; `tools/exp_issue_cost.hip` for the instruction costs):
//   matrix wave, per 48-row x 64-column tile: top 0.4 k, MFMA loop 16.9 k (480 MFMAs x 35.2; 32.0 in isolation), hand-over of the
//     accumulators + next item's rows 1.5 k, waiting for the statistics wave at the barrier 1.9 - 3.4 k: 20.8 - 22 k per tile,
//     0.43 - 0.46 k per row and tile against 0.416 k for cohort_fused2_kernel (26.6 k per 64 rows).
//   statistics wave: 21 - 22 k per tile, of which the epilogue of 48 scores per lane 15.9 k in the element-by-element form with
//     exec-masked stores, 11.8 - 16.7 k rewritten as phases of 8 independent scores with buffer stores whose offset is out of
//     range for the lanes that do not keep a score (no exec traffic at all), with and without s_setprio 3.  The same epilogue
//     takes 5.4 k cycles when the matrix waves issue no MFMAs (and no stores): beside a saturated MFMA stream the second wave
//     of a SIMD gets an instruction in about every 40 - 44 cycles, whatever their independence or its priority.
//   So the premise holds only halfway: the partner's instructions are free for the MFMA wave (32.0 cycles per MFMA beside any
//     mix of VALU, stores, exec round trips: exp_issue_cost), but the partner itself crawls — ~430 instructions per tile need
//     more than the tile's MFMA time.  cohort_fused2_kernel splits exactly this cost evenly between the two waves.
// ------------------------------------------------------------------------------------------------------------------
constexpr int kRows4 = 192;  // rows of a cohort_fused4_kernel block: 4 matrix waves x 3 row groups of 16

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

template <bool LOWEST, int NB>
__global__ __launch_bounds__(512, 1) void cohort_fused4_kernel(const FusedArgs a) {
    constexpr int NF = 4 * NB;  // 1 KiB fragments of a 64-column tile: [ks][c], lane (i16, g4) = column 16 c + i16, k 16 ks + 4 g4 ..
    constexpr int RG = 3;       // row groups of 16 per matrix wave
    constexpr int TB = 2 * NF * 64, QM = TB, P2 = QM + 32, HO = P2 + NB * 4, NX = HO + 4 * RG * 4 * 64;
    __shared__ f32x4 smem[NX + 1];
    f32x4* tbuf = smem;
    float* qms = reinterpret_cast<float*>(smem + QM);     // q_m of the two buffered tiles
    f32x4* p2s = smem + P2;                               // 2 P as fragment-shaped float4
    f32x4* hand = smem + HO;                              // [SIMD][g][c][lane]: a finished tile's accumulators
    unsigned* nxt_s = reinterpret_cast<unsigned*>(smem + NX);
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool matrix = wave8 < 4;
    const int wave = wave8 & 3;  // the SIMD, i.e. which 48 rows of the block
    const int i16 = lane & 15, g4 = lane >> 4;
    const int xcd = blockIdx.x & 7;
    const int nt64 = (int)((a.M + 63) / 64);
    const int nlb = a.nbands * a.q;
    const int per_kb = a.nfull + (a.ny - a.nfull) * a.q;
    auto lb_tile = [&](int lb) { return (int)((long long)lb * nt64 / nlb); };
    auto decode = [&](int slot, long long& rb, int& lb0, int& lbn) {  // as cohort_fused2_kernel, with blocks of kRows4 rows
        const int kb = slot / per_kb;
        const int band = kb * 8 + xcd;
        if (band >= a.nbands) return false;
        const int r = slot - kb * per_kb;
        if (r < a.nfull) {
            rb = (long long)r * kRows4;
            lb0 = band * a.q;
            lbn = lb0 + a.q;
        } else {
            const int r2 = r - a.nfull;
            rb = (long long)(a.nfull + r2 / a.q) * kRows4;
            lb0 = band * a.q + r2 % a.q;
            lbn = lb0 + 1;
        }
        return true;
    };
    auto row_of = [&](long long rb_, int g) { return rb_ + wave * (16 * RG) + 16 * g + i16; };

    long long rb = 0, nrb = 0;
    int band = 0, lbn = 0, t = 0, t1 = 0, nlb0 = 0, nlbn = 0;
    if (tid == 0) nxt_s[0] = atomicAdd(a.ctr + xcd, 1u);
    if (tid < 4 * NB) p2s[tid] = 2.0f * *reinterpret_cast<const f32x4*>(a.P + 4 * tid);
    __syncthreads();
    if (!decode(__builtin_amdgcn_readfirstlane((int)nxt_s[0]), rb, band, lbn)) return;
    t = lb_tile(band);
    t1 = lb_tile(band + 1);
    if (tid == 0) nxt_s[1] = atomicAdd(a.ctr + xcd, 1u);
    int npar = 1, buf = 0;

    if (matrix) {
        // ================================================ matrix waves ====================================================
        float brow[RG][NB][4];  // the lane's rows' operands x 2 P, in accumulation registers (defined there by v_accvgpr_write)
        // a work item's row operands, all loads of up to two row groups in flight together (a wave that serialises its loads
        // waits each of them out: there is nobody to switch to while the matrix pipe is the only thing this SIMD is for)
        auto rows_in = [&](long long rb_) {
            static_for<0, RG>([&](auto g_) {
                constexpr int g = decltype(g_)::value;
                (void)brow;  // (an asm operand alone does not make a generic lambda capture)
                long long row = row_of(rb_, g);
                if (row >= a.R) row = a.R - 1;
                const f32x4* zp = reinterpret_cast<const f32x4*>(a.zr + row * a.ldz + 4 * g4);
                f32x4 tmp[NB];
#pragma unroll
                for (int ks = 0; ks < NB; ++ks) tmp[ks] = zp[4 * ks];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int ks = 0; ks < NB; ++ks) {
                    const f32x4 v = tmp[ks] * p2s[4 * ks + g4];
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(brow[g][ks][kk]) : "v"(v[kk]));
                }
            });
        };
        auto frag_in = [&](int t_, int buf_, int f) {
            const int ks = f >> 2, c = f & 3;
            long long m = (long long)t_ * 64 + 16 * c + i16;
            if (m >= a.M) m = a.M - 1;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a.zc + m * a.ldz + 16 * ks + 4 * g4),
                                             (__attribute__((address_space(3))) void*)&tbuf[(buf_ * NF + f) * 64], 16, 0, 0);
        };
        auto qm_in = [&](int t_, int buf_) {
            long long m = (long long)t_ * 64 + lane;
            if (m >= a.M) m = a.M - 1;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a.qc + m),
                                             (__attribute__((address_space(3))) void*)&qms[buf_ * 64], 4, 0, 0);
        };
        rows_in(rb);
        for (int f = wave; f < NF; f += 4) frag_in(t, 0, f);
        if (wave == 0) qm_in(t, 0);
        __builtin_amdgcn_s_waitcnt(0x0070);  // the DMA is a pending LDS write
        __builtin_amdgcn_s_barrier();        // B1 of "tile -1": the first tile is in LDS
        unsigned long long tacc[6] = {0, 0, 0, 0, 0, 0}, tprev = __builtin_readcyclecounter();
#define ST4(k) { const unsigned long long n_ = __builtin_readcyclecounter(); tacc[k] += n_ - tprev; tprev = n_; }
        for (;;) {
            ST4(4);
            tacc[5] += 1;
            const bool last_tile = t + 1 == t1;
            const bool last_of_item = last_tile && band + 1 == lbn;
            bool have_next = true;
            int nt = t + 1;  // the next tile, for the other buffer: of this item, or the first of the next one
            if (last_of_item) {
                have_next = decode(__builtin_amdgcn_readfirstlane((int)nxt_s[npar]), nrb, nlb0, nlbn);
                nt = have_next ? lb_tile(nlb0) : -1;
            }
            // this wave's fragments of the next tile are its columns 16 wave ..: one row pointer for the tile
            const float* znext = a.zc;
            if (nt >= 0) {
                long long m = (long long)nt * 64 + 16 * wave + i16;
                if (m >= a.M) m = a.M - 1;
                znext = a.zc + m * a.ldz + 4 * g4;
            }
            f32x4 acc[RG][4];
            const f32x4* tb = tbuf + buf * NF * 64 + lane;
            f32x4 af[2][4];
#pragma unroll
            for (int c = 0; c < 4; ++c) af[0][c] = tb[c * 64];
            ST4(0);
            // k16-step ks: 48 MFMAs (kk, c, g); after the first half of the step the fragment reads of the next step and one
            // DMA piece of the next tile.  As assembly for the register files: srcB straight from the accumulation registers,
            // the accumulators in VGPRs (an allocation that parks operands in AGPRs and fetches them back pays ~66 cycles
            // per v_accvgpr_read between MFMAs); a tile's first MFMA into an accumulator starts from the constant 0.
            static_for<0, NB>([&](auto ks_) {
                constexpr int ks = decltype(ks_)::value;
                (void)brow; (void)acc; (void)af;
                static_for<0, 4>([&](auto kk_) {
                    constexpr int kk = decltype(kk_)::value;
                    (void)brow; (void)acc; (void)af;
                    static_for<0, 4 * RG>([&](auto i_) {
                        constexpr int i = decltype(i_)::value, c = i / RG, g = i % RG;
                        (void)brow; (void)acc; (void)af;
                        if ((a.abl & 16) && !(ks == 0 && kk == 0)) return;
                        if constexpr (ks == 0 && kk == 0)
                            asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, 0" : "=v"(acc[g][c]) : "v"(af[0][c][0]), "a"(brow[g][0][0]));
                        else
                            asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[g][c]) : "v"(af[ks & 1][c][kk]), "a"(brow[g][ks][kk]));
                    });
                    if constexpr (kk == 1) {
                        if constexpr (ks == 0) __builtin_amdgcn_s_barrier();  // B2: the hand-over buffer and the other tile buffer are free
                        if constexpr (ks + 1 < NB) {
#pragma unroll
                            for (int c = 0; c < 4; ++c) af[(ks + 1) & 1][c] = tb[((ks + 1) * 4 + c) * 64];
                        }
                        if (nt >= 0) {  // this wave's DMA pieces of the next tile: fragments wave + 4 ks (NF = 4 NB), q_m at the end
                            __builtin_amdgcn_global_load_lds(
                                (const __attribute__((address_space(1))) void*)(znext + 16 * ks),
                                (__attribute__((address_space(3))) void*)&tbuf[((buf ^ 1) * NF + wave + 4 * ks) * 64], 16, 0, 0);
                            if (ks == NB - 1 && wave == 0) qm_in(nt, buf ^ 1);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                });
            });
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_nop 15\n\ts_nop 7");  // the last MFMAs' results before they are read
            __builtin_amdgcn_sched_barrier(0);
            ST4(1);
            {
                f32x4* h = hand + (size_t)wave * (RG * 4 * 64) + lane;
#pragma unroll
                for (int g = 0; g < RG; ++g)
#pragma unroll
                    for (int c = 0; c < 4; ++c) h[(g * 4 + c) * 64] = acc[g][c];
            }
            if (last_of_item) {
                if (!have_next) {
                    __builtin_amdgcn_s_waitcnt(0x0070);
                    __builtin_amdgcn_s_barrier();  // B1: the statistics waves take the last tile and finish alone
                    ST4(3);
                    break;
                }
                rows_in(nrb);  // the MFMA stream of the old item is through: its operand registers are free
                rb = nrb; band = nlb0; lbn = nlbn; t = lb_tile(band); t1 = lb_tile(band + 1);
                npar ^= 1;
                if (tid == 0) nxt_s[npar] = atomicAdd(a.ctr + xcd, 1u);  // visible after B1; read >= 1 tile later
            } else if (last_tile) {
                ++band;
                ++t;
                t1 = lb_tile(band + 1);
            } else {
                ++t;
            }
            ST4(2);
            __builtin_amdgcn_s_waitcnt(0x0070);  // vmcnt(0) lgkmcnt(0): this wave's DMA pieces and its hand-over writes
            __builtin_amdgcn_s_barrier();        // B1
            ST4(3);
            buf ^= 1;
        }
        if (a.stamps && blockIdx.x == 0 && lane == 0)
            for (int i = 0; i < 6; ++i) a.stamps[wave * 6 + i] = tacc[i];
        return;
    }

    // ================================================== statistics waves ==================================================
    if (!(a.abl & 8)) __builtin_amdgcn_s_setprio(3);
    float cen[RG], thr[RG], qrv[RG], s2[RG];
    unsigned cur[RG];
    auto item_consts = [&](long long rb_) {  // per item: the rows' centre, threshold and q_r
#pragma unroll
        for (int g = 0; g < RG; ++g) {
            const long long row = row_of(rb_, g);
            const bool ok = row < a.R;
            const long long rc = ok ? row : a.R - 1;
            cen[g] = a.crow[rc];
            thr[g] = ok ? a.trow[rc] : (LOWEST ? -__builtin_inff() : __builtin_inff());  // rows past the table never append
            qrv[g] = a.qr[rc];
        }
    };
    auto band_state = [&](long long rb_, int band_) {  // per list band: an empty sum and the sub-lists' first slots
#pragma unroll
        for (int g = 0; g < RG; ++g) {
            long long rc = row_of(rb_, g);
            if (rc >= a.R) rc = a.R - 1;
            s2[g] = 0.f;
            cur[g] = 4u * (unsigned)(((rc * nlb + band_) * a.ksub) * 4 + g4);
        }
    };
    auto band_end = [&](long long rb_, int band_) {
#pragma unroll
        for (int g = 0; g < RG; ++g) {
            const long long row = row_of(rb_, g);
            double u2 = (double)s2[g];  // the four lane groups of a row: ((g0 + g1) + (g2 + g3)), as cohort_fused2_kernel
            u2 += __hiloint2double(__shfl_xor(__double2hiint(u2), 16, 64), __shfl_xor(__double2loint(u2), 16, 64));
            u2 += __hiloint2double(__shfl_xor(__double2hiint(u2), 32, 64), __shfl_xor(__double2loint(u2), 32, 64));
            if (row < a.R) {
                const unsigned sidx = (unsigned)(band_ * 4 + g4);
                a.counts[(size_t)row * a.nsub + sidx] = (cur[g] / 4u - (unsigned)(((row * nlb + band_) * a.ksub) * 4 + g4)) / 4u;
                if (g4 == 0) a.part[(size_t)row * (a.nsub / 4) + band_] = u2;
            }
        }
    };
    const unsigned stride_b = 16u;  // bytes between consecutive slots of a sub-list
    // the candidate lists as a raw buffer: offsets at or past its size are dropped (max_rows keeps it below 2^30 bytes)
    const __amdgpu_buffer_rsrc_t lres = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.lists), 0, (unsigned)((size_t)a.R * a.nsub * a.ksub * 4), 0x00020000);
    // the statistics of one finished tile: the copy of its accumulators and q_m, its rows (prb), list band (pband), first column
    // (pm0); same arithmetic and order per row group as cohort_fused2_kernel's epilogue
    f32x4 accp[RG][4], qmp[4];
    auto epilogue = [&](auto masked_, long long prb, int pband, long long pm0) {
        constexpr bool MASKED = decltype(masked_)::value;
        typedef float f32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int g = 0; g < RG; ++g) {
            f32x2 pq2 = {s2[g], 0.f};
            unsigned o = cur[g];
            const float th = thr[g];
            // Eight scores (two accumulator blocks) at a time, in PHASES of independent instructions: this wave gets an issue
            // slot now and then beside the matrix wave (one dependent instruction per ~44 cycles, measured: 15.9 k cycles per
            // tile for the element-by-element form), so what it needs is several instructions ready at once.
            //   if (s <= th) { lists[o] = s; o += stride; }   (>= for the N largest)
            // without touching exec: a buffer store whose offset is out of range is dropped by the address unit, so a lane that
            // does not keep a score stores it to offset ~0; the cursor steps are a short prefix sum.
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                float sv[8], dv[8];
#pragma unroll
                for (int cc = 0; cc < 2; ++cc) {
                    const int c = 2 * h + cc;
                    f32x4 s4 = accp[g][c] + (qmp[c] + qrv[g]);  // the score, same bits as the spilling kernel
                    f32x4 d4 = s4 - cen[g];                      // centred on the row's analytic mean
                    if (MASKED) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const bool okc = pm0 + 16 * c + 4 * g4 + r < a.M;
                            d4[r] = okc ? d4[r] : 0.f;
                            s4[r] = okc ? s4[r] : (LOWEST ? __builtin_inff() : -__builtin_inff());
                        }
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) { sv[4 * cc + r] = s4[r]; dv[4 * cc + r] = d4[r]; }
                }
                __builtin_amdgcn_sched_barrier(0);
                unsigned inc[8], nk[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const bool keep = LOWEST ? sv[e] <= th : sv[e] >= th;
                    inc[e] = keep ? stride_b : 0u;
                    nk[e] = keep ? 0u : 0xffffffffu;
                }
                __builtin_amdgcn_sched_barrier(0);
                // cursor of element e: o + (steps of the elements before it); two chains of four, then the second one's base
                unsigned off[8];
                off[0] = o;            off[4] = 0u;
                off[1] = o + inc[0];   off[5] = inc[4];
                off[2] = off[1] + inc[1]; off[6] = off[5] + inc[5];
                off[3] = off[2] + inc[2]; off[7] = off[6] + inc[6];
                const unsigned mid = off[3] + inc[3], tot1 = off[7] + inc[7];
#pragma unroll
                for (int e = 4; e < 8; ++e) off[e] += mid;
                o = mid + tot1;
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (!(a.abl & 32)) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(sv[e]), lres, off[e] | nk[e], 0, 0);
                    else asm volatile("" :: "v"(sv[e]), "v"(off[e] | nk[e]));
                // the squares, in cohort_fused2_kernel's order (pairs (0, 1) and (2, 3) of each block into the two chains)
#pragma unroll
                for (int cc = 0; cc < 2; ++cc) {
                    const f32x2 dl = {dv[4 * cc], dv[4 * cc + 1]}, dh = {dv[4 * cc + 2], dv[4 * cc + 3]};
                    pq2 = __builtin_elementwise_fma(dl, dl, pq2);
                    pq2 = __builtin_elementwise_fma(dh, dh, pq2);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            s2[g] = pq2[0] + pq2[1];
            // at most ksub - kSubSlack entries stay (the select kernel treats that count as an overflow)
            long long rc = row_of(prb, g);
            if (rc >= a.R) rc = a.R - 1;
            const unsigned lim = 4u * (unsigned)(((rc * nlb + pband) * a.ksub + (a.ksub - kSubSlack)) * 4 + g4);
            cur[g] = o < lim ? o : lim;
        }
    };
    // the tile whose accumulators are in the hand-over buffer after the next B1
    bool pending = false, p_first_of_item = false, p_first_of_band = false, p_last_of_band = false;
    long long prb = 0;
    int pband = 0, pt = 0, pbuf = 0;
    bool first_of_item = true, first_of_band = true;
    unsigned long long cacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, cprev = __builtin_readcyclecounter();
#define SC4(k) { const unsigned long long n_ = __builtin_readcyclecounter(); cacc[k] += n_ - cprev; cprev = n_; }
    auto take_and_run = [&](bool with_b2) {
        SC4(0);
        {   // the copy: accumulators from the hand-over buffer, q_m from the tile's buffer
            const f32x4* h = hand + (size_t)wave * (RG * 4 * 64) + lane;
#pragma unroll
            for (int g = 0; g < RG; ++g)
#pragma unroll
                for (int c = 0; c < 4; ++c) accp[g][c] = h[(g * 4 + c) * 64];
            const float* qm_s = qms + pbuf * 64 + 4 * g4;
#pragma unroll
            for (int c = 0; c < 4; ++c) qmp[c] = *reinterpret_cast<const f32x4*>(qm_s + 16 * c);
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);              // lgkmcnt(0)
        SC4(1);
        if (with_b2) __builtin_amdgcn_s_barrier();       // B2
        SC4(2);
        if (p_first_of_item) item_consts(prb);
        if (p_first_of_band) band_state(prb, pband);
        SC4(3);
        const long long pm0 = (long long)pt * 64;
        if (pm0 + 64 > a.M) epilogue(std::true_type{}, prb, pband, pm0);
        else epilogue(std::false_type{}, prb, pband, pm0);
        SC4(4);
        if (p_last_of_band) band_end(prb, pband);
        SC4(5);
        cacc[7] += 1;
    };
    __builtin_amdgcn_s_barrier();  // B1 of "tile -1"
    for (;;) {
        const bool last_tile = t + 1 == t1;
        const bool last_of_item = last_tile && band + 1 == lbn;
        bool have_next = true;
        if (last_of_item) have_next = decode(__builtin_amdgcn_readfirstlane((int)nxt_s[npar]), nrb, nlb0, nlbn);
        // while the matrix waves work on this tile: the statistics of the one before
        if (pending) take_and_run(true);
        else __builtin_amdgcn_s_barrier();  // B2
        // this tile becomes the pending one
        pending = true;
        prb = rb; pband = band; pt = t; pbuf = buf;
        p_first_of_item = first_of_item; p_first_of_band = first_of_band; p_last_of_band = last_tile;
        first_of_item = false;
        first_of_band = last_tile;
        if (last_of_item) {
            if (!have_next) {
                __builtin_amdgcn_s_barrier();  // B1: the last tile's accumulators are there
                take_and_run(false);
                break;
            }
            rb = nrb; band = nlb0; lbn = nlbn; t = lb_tile(band); t1 = lb_tile(band + 1);
            npar ^= 1;
            first_of_item = true;
        } else if (last_tile) {
            ++band;
            ++t;
            t1 = lb_tile(band + 1);
        } else {
            ++t;
        }
        SC4(0);
        __builtin_amdgcn_s_barrier();  // B1
        SC4(6);
        buf ^= 1;
    }
    if (a.stamps && blockIdx.x == 0 && lane == 0)
        for (int i = 0; i < 8; ++i) a.stamps[32 + wave * 8 + i] = cacc[i];
}

