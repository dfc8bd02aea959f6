// tools/exp_cohort_fused3_kernel.h — EXPERIMENT, part of no build.  Round 3's sixth attempt at the AS-norm statistics kernel
// (neuralplda_amd/csrc/nplda_cohort_fused.hip, cohort_fused2_kernel): the same work items, tiles, lists and results, with ONE
// wave per SIMD instead of two.  It passes tests/test_cohort_fused_gpu.py + tests/test_asnorm_gpu.py (44 tests) when dropped in
// behind cohort_fused2_kernel (it uses that file's FusedArgs, f32x4, kSubSlack; launch with dim3(256); NPLDA_CF3_STAMP adds the
// cycle stamps the numbers below come from, FusedArgs then needs an `unsigned long long* stamps`), and it is NOT faster:
//
//   cfg3 (R = 22000, M = 10000, D = 150), same box, rocprofv3 --kernel-trace: 716 us against 704 us for cohort_fused2_kernel
//   (other boxes: 733 / 677).  Cycles per 64-column tile of one wave (s_memtime, block 0):
//       MFMA loop, no epilogue riding in it      21.3k   = 640 MFMAs x 33.3: the matrix pipe is full, which two waves never reach
//       MFMA loop with the previous tile's epilogue as 128 slots between the MFMAs   24.7k   (+26 cycles per slot)
//       after the loop (accumulators aside; per item: rows, epilogue of the last tile, counts, sums)   2.7-4.0k
//       top of tile + barrier                     0.7-1.6k
//   i.e. 28-30k per tile against 26.6k for the two-wave kernel.  What the experiment established:
//   * v_accvgpr_read beside MFMAs costs ~66 cycles apiece.  With 512 registers per wave the allocator parks row operands in the
//     accumulation registers and fetches them back (65-73 reads per tile, + 64 to move the accumulators out): 26.9k -> 21.4k per
//     loop once the MFMAs were written as assembly with the row operands defined IN AGPRs ("a" constraint, read as srcB directly)
//     and the accumulators in ordinary VGPRs.  cohort_fused2_kernel's loop has no such reads (checked in its assembly).
//   * A wave alone on its SIMD does not hide VALU / SALU work "in the shadow" of its own MFMAs: every epilogue instruction put
//     between MFMAs costs about its issue time (5 cycles; 26 per slot of 5), an exec written by v_cmpx costs 80.  The second wave
//     of the shipped kernel is what hides the epilogue.
//   * Serialised loads are fully exposed: the next item's 40 row-operand loads, issued one at a time for lack of registers, cost
//     28 us per item until they were batched.
// ------------------------------------------------------------------------------------------------------------------
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

template <bool LOWEST, int NB>
__global__ __launch_bounds__(256, 1) void cohort_fused3_kernel(const FusedArgs a) {
    constexpr int NF = 4 * NB;  // 1 KiB fragments of a 64-column tile: [ks][c], lane (i16, g4) = column 16 c + i16, k 16 ks + 4 g4 ..
    constexpr int NW = 4, RG = 4;  // waves, row groups of 16 per wave
    __shared__ f32x4 smem[2 * NF * 64 + 2 * 16 + NB * 4 + 2];
    f32x4* tbuf = smem;
    float* qms = reinterpret_cast<float*>(smem + 2 * NF * 64);                 // q_m of the two buffered tiles
    f32x4* p2s = smem + 2 * NF * 64 + 32;                                      // 2 P as fragment-shaped float4
    unsigned* nxt_s = reinterpret_cast<unsigned*>(smem + 2 * NF * 64 + 32 + NB * 4);
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i16 = lane & 15, g4 = lane >> 4;
    const int xcd = blockIdx.x & 7;
    const int nt64 = (int)((a.M + 63) / 64);
    const int nlb = a.nbands * a.q;
    const int per_kb = a.nfull + (a.ny - a.nfull) * a.q;
    auto lb_tile = [&](int lb) { return (int)((long long)lb * nt64 / nlb); };
    auto decode = [&](int slot, long long& rb, int& lb0, int& lbn) {  // as cohort_fused2_kernel
        const int kb = slot / per_kb;
        const int band = kb * 8 + xcd;
        if (band >= a.nbands) return false;
        const int r = slot - kb * per_kb;
        if (r < a.nfull) {
            rb = (long long)r * 256;
            lb0 = band * a.q;
            lbn = lb0 + a.q;
        } else {
            const int r2 = r - a.nfull;
            rb = (long long)(a.nfull + r2 / a.q) * 256;
            lb0 = band * a.q + r2 % a.q;
            lbn = lb0 + 1;
        }
        return true;
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

    long long rb = 0, nrb = 0;
    int band = 0, lbn = 0, t = 0, t1 = 0, nlb0 = 0, nlbn = 0;
    if (tid == 0) nxt_s[0] = atomicAdd(a.ctr + xcd, 1u);
    if (tid < 4 * NB) p2s[tid] = 2.0f * *reinterpret_cast<const f32x4*>(a.P + 4 * tid);
    __syncthreads();
    if (!decode(__builtin_amdgcn_readfirstlane((int)nxt_s[0]), rb, band, lbn)) return;
    t = lb_tile(band);
    t1 = lb_tile(band + 1);
    if (tid == 0) nxt_s[1] = atomicAdd(a.ctr + xcd, 1u);
    int npar = 1;

    // per-lane state of a work item: the lane's four rows (row group g, row i16 of it) and their operand fragments
    float brow[RG][NB][4];  // in accumulation registers from item_rows on (defined there, by v_accvgpr_write, as such)
    float cen[RG], thr[RG], qrv[RG], s2[RG];
    unsigned cur[RG];
    auto row_of = [&](long long rb_, int g) { return rb_ + wave * 64 + 16 * g + i16; };
    // a work item's row operands, two row groups at a time: all of their loads in flight together (a wave alone on its SIMD
    // waits out every load it serialises — 28 us per item when the allocator, short of registers, took them one by one)
    auto rows_load = [&](long long rb_, int g0, f32x4 (&tmp)[2][NB]) {
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            long long row = row_of(rb_, g0 + g);
            if (row >= a.R) row = a.R - 1;
            const f32x4* zp = reinterpret_cast<const f32x4*>(a.zr + row * a.ldz + 4 * g4);
#pragma unroll
            for (int ks = 0; ks < NB; ++ks) tmp[g][ks] = zp[4 * ks];
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    auto rows_put = [&](int g0, f32x4 (&tmp)[2][NB]) {  // times 2 P, into the accumulation registers
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int ks = 0; ks < NB; ++ks) {
                const f32x4 v = tmp[g][ks] * p2s[4 * ks + g4];
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    if (g0 == 0) asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(brow[g][ks][kk]) : "v"(v[kk]));
                    else asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(brow[2 + g][ks][kk]) : "v"(v[kk]));
                }
            }
    };
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
    auto item_end = [&](long long rb_, int band_) {
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

    {
        f32x4 tmp[2][NB];
        rows_load(rb, 0, tmp);
        rows_put(0, tmp);
        rows_load(rb, 2, tmp);
        rows_put(2, tmp);
    }
    item_consts(rb);
    band_state(rb, band);
    for (int f = wave; f < NF; f += NW) frag_in(t, 0, f);
    if (wave == 0) qm_in(t, 0);
    __syncthreads();  // (drains the DMA: it is a pending LDS write)
    int buf = 0;
    const unsigned stride_b = 16u;  // bytes between consecutive slots of a sub-list
    const float* lbase = a.lists;

    // the previous tile's accumulators / q_m / column base while its epilogue is pending, and the epilogue's running sums
    f32x4 accp[RG][4], qmp[4];
    float sq2[RG][2];
    bool pending = false, plast = false;  // plast: the pending tile is the last of its list band pband
    int pband = 0;
#pragma unroll
    for (int g = 0; g < RG; ++g)
#pragma unroll
        for (int c = 0; c < 4; ++c) accp[g][c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 4; ++c) qmp[c] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- the epilogue in pieces (same arithmetic, same order per row group as cohort_fused2_kernel's) --------------------
    auto epi_begin = [&]() {
#pragma unroll
        for (int g = 0; g < RG; ++g) { sq2[g][0] = s2[g]; sq2[g][1] = 0.f; }
    };
    auto epi_sq = [&](const f32x4 accv, const f32x4 qmv, int g, bool masked, long long m0_, int c, f32x4& s4) {
        s4 = accv + (qmv + qrv[g]);          // the score, same bits as the spilling kernel
        f32x4 d4 = s4 - cen[g];              // centred on the row's analytic mean
        if (masked) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool okc = m0_ + 16 * c + 4 * g4 + r < a.M;
                d4[r] = okc ? d4[r] : 0.f;
                s4[r] = okc ? s4[r] : (LOWEST ? __builtin_inff() : -__builtin_inff());
            }
        }
        // two chains, as lanes 0 / 1 of the packed fma of cohort_fused2_kernel
        sq2[g][0] = fmaf(d4[0], d4[0], sq2[g][0]);
        sq2[g][1] = fmaf(d4[1], d4[1], sq2[g][1]);
        sq2[g][0] = fmaf(d4[2], d4[2], sq2[g][0]);
        sq2[g][1] = fmaf(d4[3], d4[3], sq2[g][1]);
    };
    // if (s <= th) { lists[o] = s; o += stride; }  (>= for the N largest) as an exec-masked store: no branch, SGPR base + 32-bit
    // byte offset, one VALU for the cursor
    auto epi_app = [&](float sv_, int g) {
        unsigned long long sv, k0;
        unsigned o = cur[g];
        const float th = thr[g];
#define NPLDA_APPEND1(CMP)                                                                                             \
    asm volatile(CMP " %[k0], %[s0], %[th]\n\t"                                                                        \
                 "s_mov_b64 %[sv], exec\n\t"                                                                           \
                 "s_mov_b64 exec, %[k0]\n\tglobal_store_dword %[o], %[s0], %[base]\n\tv_add_u32 %[o], %[o], %[st]\n\t" \
                 "s_mov_b64 exec, %[sv]"                                                                                \
                 : [o] "+v"(o), [sv] "=&s"(sv), [k0] "=&s"(k0)                                                         \
                 : [th] "v"(th), [s0] "v"(sv_), [base] "s"(lbase), [st] "s"(stride_b)                                   \
                 : "memory")
        if (LOWEST) NPLDA_APPEND1("v_cmp_le_f32");
        else NPLDA_APPEND1("v_cmp_ge_f32");
#undef NPLDA_APPEND1
        cur[g] = o;
    };
    auto epi_end = [&](long long rb_, int band_) {
#pragma unroll
        for (int g = 0; g < RG; ++g) {
            s2[g] = sq2[g][0] + sq2[g][1];
            // at most ksub - kSubSlack entries stay (the select kernel treats that count as an overflow)
            long long rc = row_of(rb_, g);
            if (rc >= a.R) rc = a.R - 1;
            const unsigned lim = 4u * (unsigned)(((rc * nlb + band_) * a.ksub + (a.ksub - kSubSlack)) * 4 + g4);
            cur[g] = cur[g] < lim ? cur[g] : lim;
        }
    };

#ifdef NPLDA_CF3_STAMP
    // acc_t[5 d + k]: cycles from stamp k to stamp k + 1 (4 -> 0 of the next tile) over tiles with (d = 1) / without a pending
    // epilogue; acc_t[10 + d]: tiles
    unsigned long long acc_t[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, t_prev = __builtin_readcyclecounter();
    int dsel = 0;
#define NPLDA_STAMP(k)                                                          \
    {                                                                           \
        if ((k) == 0) { dsel = pending ? 1 : 0; acc_t[10 + dsel] += 1; }        \
        const unsigned long long now_ = __builtin_readcyclecounter();           \
        acc_t[5 * dsel + ((k) + 4) % 5] += now_ - t_prev;                       \
        t_prev = now_;                                                          \
    }
#else
#define NPLDA_STAMP(k)
#endif
    for (;;) {
        const bool last_tile = t + 1 == t1;                       // of the list band
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

        // k16-step ks: 64 MFMAs in 8 groups of 8 (kk, half of the columns: 2 c x 4 g); behind group q of the step comes piece
        // p = 8 ks + q of the pending epilogue (80 pieces = 16 blocks (g, c) x {squares, 4 appends}; NB < 10 packs what is
        // left behind the last group), and behind group 3 the fragment reads of the next step and one DMA piece
        auto mfma_loop = [&](auto deferred) {
            constexpr bool DEF = decltype(deferred)::value;
            constexpr int NM = 64 * NB, NSLOT = 128;  // MFMAs of the tile; slots of the pending epilogue (16 blocks x 4 r x 2)
            float sp = 0.f;
            unsigned long long keep = 0;
            if (DEF) epi_begin();
            // slot 2 e: the score of element e = (g, c, r) of the previous tile, its centred square and the comparison with
            // the row's threshold; slot 2 e + 1: the append under that comparison.  (One slot per element with v_cmpx writing
            // exec directly was SLOWER: 81 cycles per element instead of 2 x 26 — a VALU write of exec stalls what follows.)
            auto slot = [&](auto p_) {
                constexpr int p = decltype(p_)::value, e = p >> 1, u = e >> 2, r = e & 3, g = u >> 2, c = u & 3;
                if constexpr ((p & 1) == 0) {
                    sp = accp[g][c][r] + (qmp[c][r] + qrv[g]);
                    const float d = sp - cen[g];
                    sq2[g][r & 1] = fmaf(d, d, sq2[g][r & 1]);
                    asm volatile("" : "+v"(sq2[g][r & 1]));  // here, not sunk behind the loop with 64 scores kept alive for it
                    if (LOWEST) asm volatile("v_cmp_le_f32 %0, %1, %2" : "=s"(keep) : "v"(sp), "v"(thr[g]));
                    else asm volatile("v_cmp_ge_f32 %0, %1, %2" : "=s"(keep) : "v"(sp), "v"(thr[g]));
                } else {
                    unsigned long long sv;
                    unsigned o = cur[g];
                    asm volatile("s_mov_b64 %[sv], exec\n\t"
                                 "s_mov_b64 exec, %[k0]\n\tglobal_store_dword %[o], %[s0], %[base]\n\tv_add_u32 %[o], %[o], %[st]\n\t"
                                 "s_mov_b64 exec, %[sv]"
                                 : [o] "+v"(o), [sv] "=&s"(sv)
                                 : [k0] "s"(keep), [s0] "v"(sp), [base] "s"(lbase), [st] "s"(stride_b)
                                 : "memory");
                    cur[g] = o;
                }
            };
            static_for<0, NB>([&](auto ks_) {
                constexpr int ks = decltype(ks_)::value;
                static_for<0, 8>([&](auto q_) {
                    constexpr int q = decltype(q_)::value, kk = q >> 1, c0 = 2 * (q & 1);
                    static_for<0, 8>([&](auto i_) {
                        constexpr int i = decltype(i_)::value, c = c0 + i / RG, g = i % RG;
                        // as assembly for the register files: the rows' operands stay in the accumulation registers for the whole
                        // item and are read from there (the compiler's own allocation parked some of them there and fetched them
                        // with v_accvgpr_read — 66 cycles apiece beside MFMAs, measured), the accumulators live in ordinary
                        // VGPRs where the epilogue reads them
                        // (a tile's first MFMA into an accumulator starts from the constant 0: nothing to initialise)
                        if constexpr (ks == 0 && kk == 0)
                            asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, 0" : "=v"(acc[g][c]) : "v"(af[0][c][0]), "a"(brow[g][0][0]));
                        else
                            asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[g][c]) : "v"(af[ks & 1][c][kk]), "a"(brow[g][ks][kk]));
                        if constexpr (DEF) {
                            constexpr int jm = (8 * ks + q) * 8 + i;  // index of this MFMA in the tile
                            // the slots spread over the first 4 / 5 of the tile: the appends have landed by its end
                            constexpr int NME = NM * 4 / 5;
                            constexpr int p0 = jm * NSLOT / NME < NSLOT ? jm * NSLOT / NME : NSLOT;
                            constexpr int p1 = (jm + 1) * NSLOT / NME < NSLOT ? (jm + 1) * NSLOT / NME : NSLOT;
                            if constexpr (p1 > p0) {
                                __builtin_amdgcn_sched_barrier(0);
                                static_for<p0, p1>([&](auto p_) { slot(p_); });
                                __builtin_amdgcn_sched_barrier(0);
                            }
                        }
                    });
                    if constexpr (q == 3) {
                        if constexpr (ks + 1 < NB) {
#pragma unroll
                            for (int c = 0; c < 4; ++c) af[(ks + 1) & 1][c] = tb[((ks + 1) * 4 + c) * 64];
                        }
                        if (nt >= 0) {  // this wave's DMA pieces of the next tile: fragments wave + 4 ks (NF = 4 NB), q_m at the end
                            __builtin_amdgcn_global_load_lds(
                                (const __attribute__((address_space(1))) void*)(znext + 16 * ks),
                                (__attribute__((address_space(3))) void*)&tbuf[((buf ^ 1) * NF + wave + NW * ks) * 64], 16, 0, 0);
                            if (ks == NB - 1 && wave == 0) qm_in(nt, buf ^ 1);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                });
            });
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_nop 15\n\ts_nop 7");  // the last MFMAs' results before the VALU reads them
            __builtin_amdgcn_sched_barrier(0);
            if (DEF) epi_end(rb, pband);
        };
        NPLDA_STAMP(0);
        if (pending) {
            mfma_loop(std::true_type{});
            if (plast) {  // the pending tile closed its list band: the band's counts and sum, then this tile's band
                item_end(rb, pband);
                band_state(rb, band);
            }
        } else {
            mfma_loop(std::false_type{});
        }
        NPLDA_STAMP(1);
        // this wave's part of the next tile has landed (and the pending tile's appends, 1 / 5 of a tile old, are out); what
        // is stored from here on (a band's counts and sum, the appends of an item's last tile) is not waited for
        __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0)
        NPLDA_STAMP(2);
        // the operand rows of the NEXT item are fetched under the tail of this item's last tile: the first half's loads fly
        // during its epilogue, the second half's during the bookkeeping of the item switch
        f32x4 rtmp[2][NB];
        if (last_of_item && have_next) rows_load(nrb, 0, rtmp);

        const long long m0 = (long long)t * 64;
        const float* qm_s = qms + buf * 64 + 4 * g4;
        f32x4 qv[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) qv[c] = *reinterpret_cast<const f32x4*>(qm_s + 16 * c);
        if (!last_of_item) {
            // this tile's epilogue rides in the next tile's loop (of the same rows): its accumulators and q_m move aside (the
            // tile's LDS buffer, q_m included, is the DMA target of the tile after next)
#pragma unroll
            for (int g = 0; g < RG; ++g)
#pragma unroll
                for (int c = 0; c < 4; ++c) accp[g][c] = acc[g][c];
#pragma unroll
            for (int c = 0; c < 4; ++c) qmp[c] = qv[c];
            pending = true;
            pband = band;
            plast = last_tile;
        } else {
            // last tile of the item (the only one that can reach past column M): its epilogue at once
            const bool masked = m0 + 64 > a.M;
            epi_begin();
#pragma unroll
            for (int g = 0; g < RG; ++g)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    f32x4 s4;
                    epi_sq(acc[g][c], qv[c], g, masked, m0, c, s4);
#pragma unroll
                    for (int r = 0; r < 4; ++r) epi_app(s4[r], g);
                }
            epi_end(rb, band);
            pending = false;
            if (have_next) {
                rows_put(0, rtmp);
                rows_load(nrb, 2, rtmp);
            }
        }

        if (last_tile) {
            if (!last_of_item) {  // next list band of the same item: same rows, the tiles go on; the band's state changes
                ++band;           // hands when its last tile's epilogue is through (above)
                ++t;
                t1 = lb_tile(band + 1);
            } else {
                item_end(rb, band);
                if (!have_next) break;
                rb = nrb; band = nlb0; lbn = nlbn; t = lb_tile(band); t1 = lb_tile(band + 1);
                item_consts(rb);
                band_state(rb, band);
                rows_put(2, rtmp);
                npar ^= 1;
                if (tid == 0) nxt_s[npar] = atomicAdd(a.ctr + xcd, 1u);  // visible after the barrier below; read >= 1 tile later
            }
        } else {
            ++t;
        }
        // end of tile: everybody's part of the next tile has landed; nobody still reads the buffer the tile after next
        // will overwrite
        NPLDA_STAMP(3);
        __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
        __builtin_amdgcn_s_barrier();
        NPLDA_STAMP(4);
        buf ^= 1;
    }
#ifdef NPLDA_CF3_STAMP
    if (blockIdx.x == 0 && lane == 0)
        for (int i = 0; i < 12; ++i) a.stamps[wave * 12 + i] = acc_t[i];
#endif
}

