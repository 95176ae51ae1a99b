// k_aclist.hip -- progressive AC scans coded from COMPACTED COEFFICIENT LISTS (first-pass scans: Ah = 0).
//
// Replaces, for those scans, the per-block sweeps of k_entropy.hip's k_tokens and the token stream between it and k_pack (mozjpeg
// jcphuff.c encode_mcu_AC_first + jchuff.c statistics, reached from /root/reference/src/compressor.rs:305).
//
// Why: the scan search codes ~30 first-pass candidates per file, and a lane that owns a block spends a step on each of its 63
// positions in every one of them although one coefficient in eight is non-zero.  Here the non-zero coefficients of a component are
// written down ONCE per point transform Al -- an NzList (types.h): per block its entries in zig-zag order, then an END entry -- and every
// candidate scan (Ss, Se, Al) is a FLAT walk over that list, one entry per lane:
//   * an entry with Ss <= k <= Se is a coded coefficient; its zero run is the distance to the entry in front of it when that one belongs
//     to the same block and band (the neighbouring lane: lists are sorted), else to the band's start;
//   * the first entry of a block that lies behind the band (k > Se; the END entry at the latest) stands for the block's end: it knows
//     whether the block coded anything (has-symbol) and whether it ends with an EOB -- the two flags k_ac_runs builds the EOB runs from --
//     and, in the pack pass, emits the EOBRUN symbol the block owns.
// No token is written: k_list_stats takes the histograms (and the flags), k_list_pack derives the same events again and turns them into bits
// with the optimal tables -- an event is a dozen instructions, a token was a 4-byte store and a 4-byte load.
// The kernels are written once for both builds (wave.h): the emulation runs the statements the device runs.
#include "kernels.h"
#include "wave.h"

namespace csh {

#define CSH_LP_WORDS 1024   // the packer's window of the bit stream, per wave, in LDS words (a step of 256 entries adds at most 256 x 79 bits = 632 words)

__device__ __forceinline__ static uint32_t lbitlen(uint32_t v) { return 32u - uint32_t(__clz(v)); }
__device__ __forceinline__ static int lwave() { return int(threadIdx.x) / CSP_WAVE_THREADS; }

// ------------------------------------------------------------------------------------------------ the builder
// Level 0 (k_nzlist): one wave per 256-block chunk of a component.  Lane (b, o) = (l >> 3, l & 7) holds octet o (coefficients 8 o .. 8 o + 7:
// one 16-byte load) of block 8 s + b in step s, so the lanes of a step, in lane order, hold 8 blocks' coefficients in list order: a wave
// scan of the per-lane counts places every lane's entries, and the lanes' stores land next to each other.  A wave takes half a chunk (16
// steps: 16 loads in flight per lane, 64 registers) and both passes -- the entry count, then, behind one atomic add on the list's cursor,
// the entries -- run from those registers: the kernel waits for memory once.
// The other levels (k_nzfilter): flat over the level-0 chunk, entry -> |c| >> Al, dropped when that is zero.
#define CSH_NZ_HALF 16   // steps of 8 blocks held in registers at a time: half a chunk
__device__ __forceinline__ static uint4 nz_load(const EncCtx &c, const NzSet &S, uint32_t u, int bx, int by, uint32_t oct) {
    uint4 q; q.x = q.y = q.z = q.w = 0u;
    if (u < S.nunits) {
        const int b = by * S.bw + bx;
        q = *reinterpret_cast<const uint4 *>(c.coef + (size_t(S.tile_base) + size_t(b >> 6)) * CSH_TILE_I16 + size_t((b & 63) * CSH_BLK_STRIDE) + size_t(oct) * CSH_OCT_STRIDE);
    }
    return q;
}
// bit i: coefficient i of the octet is a non-zero AC coefficient
__device__ __forceinline__ static uint32_t nz_mask8(const uint4 &q, uint32_t oct) {
    uint32_t m = 0;
    m |= (q.x & 0xFFFFu) ? 1u : 0u;   m |= (q.x >> 16) ? 2u : 0u;
    m |= (q.y & 0xFFFFu) ? 4u : 0u;   m |= (q.y >> 16) ? 8u : 0u;
    m |= (q.z & 0xFFFFu) ? 16u : 0u;  m |= (q.z >> 16) ? 32u : 0u;
    m |= (q.w & 0xFFFFu) ? 64u : 0u;  m |= (q.w >> 16) ? 128u : 0u;
    return oct ? m : (m & ~1u);
}
// half `half` (128 blocks, 16 steps) of the chunk that starts at block u0: every lane's 16 octets, all loads in flight at once
__device__ __forceinline__ static void nz_load_half(const EncCtx &c, const NzSet &S, uint32_t u0, int half, LV<uint4> (&q)[CSH_NZ_HALF]) {
    LFOR(l) {
        const uint32_t oct = uint32_t(l & 7);
        uint32_t u = u0 + 128u * uint32_t(half) + uint32_t(l >> 3);
        int by = int(u) / S.real_bw, bx = int(u) - by * S.real_bw;
        CSH_UNROLL
        for (int s = 0; s < CSH_NZ_HALF; s++) {
            q[s][l] = nz_load(c, S, u, bx, by, oct);
            u += 8u; bx += 8;
            while (bx >= S.real_bw) { bx -= S.real_bw; by++; }   // eight blocks on
        }
    }
}
// entries of the half per lane: its non-zero AC coefficients; the lane of octet 7 adds the block's END
__device__ __forceinline__ static void nz_count_half(const NzSet &S, uint32_t u0, int half, const LV<uint4> (&q)[CSH_NZ_HALF], LV<uint32_t> &cnt) {
    LFOR(l) {
        const uint32_t oct = uint32_t(l & 7);
        uint32_t n = 0;
        CSH_UNROLL
        for (int s = 0; s < CSH_NZ_HALF; s++)
            n += uint32_t(__popc(nz_mask8(q[s][l], oct))) + ((oct == 7u && u0 + 128u * uint32_t(half) + 8u * uint32_t(s) + uint32_t(l >> 3) < S.nunits) ? 1u : 0u);
        cnt[l] += n;
    }
}
__device__ __forceinline__ static void nz_write_half(const NzSet &S, uint32_t u0, int half, const LV<uint4> (&q)[CSH_NZ_HALF], uint32_t *dst, uint32_t run, uint8_t *blk_cnt, uint16_t *blk_off) {
    CSH_UNROLL
    for (int s = 0; s < CSH_NZ_HALF; s++) {
        LV<uint32_t> mk, cl;
        LFOR(l) {
            const uint32_t oct = uint32_t(l & 7);
            mk[l] = nz_mask8(q[s][l], oct);
            cl[l] = uint32_t(__popc(mk[l])) + ((oct == 7u && u0 + 128u * uint32_t(half) + 8u * uint32_t(s) + uint32_t(l >> 3) < S.nunits) ? 1u : 0u);
        }
        uint32_t tot;
        const LV<uint32_t> ex = lscan(cl, tot);
        if (blk_cnt) {   // the block's entries = where its END lands - where its first octet starts (seven lanes down)
            LV<uint32_t> first = ex;
            for (int d = 0; d < 7; d++) first = lprev(first, 0u);
            LFOR(l) {
                const uint32_t blk = 16u * 8u * uint32_t(half) + 8u * uint32_t(s) + uint32_t(l >> 3);
                if ((l & 7) == 7 && u0 + blk < S.nunits) blk_cnt[u0 + blk] = uint8_t(ex[l] + cl[l] - 1u - first[l]);
                if ((l & 7) == 7 && u0 + blk < S.nunits && blk_off) blk_off[u0 + blk] = uint16_t(run + first[l]);   // (a chunk holds at most 256 x 64 entries)
            }
        }
        LFOR(l) {
            const uint32_t oct = uint32_t(l & 7);
            const uint32_t blk = 16u * 8u * uint32_t(half) + 8u * uint32_t(s) + uint32_t(l >> 3);
            const uint32_t base = (8u * oct) | (blk << 23);
            const uint32_t w[4] = {q[s][l].x, q[s][l].y, q[s][l].z, q[s][l].w};
            uint32_t o = run + ex[l];
            CSH_UNROLL
            for (int i = 0; i < 8; i++)
                if ((mk[l] >> i) & 1u) {
                    const int h = (i & 1) ? (int(w[i >> 1]) >> 16) : (int(w[i >> 1] << 16) >> 16);
                    const uint32_t a = h == -32768 ? 32767u : uint32_t(h < 0 ? -h : h);   // 15 bits of magnitude: -32768 (a crafted progressive input; no encoder's coefficient) must not reach the block field
                    dst[o++] = (base + uint32_t(i)) | (h < 0 ? 128u : 0u) | (a << 8);
                }
            if (oct == 7u && u0 + blk < S.nunits) dst[o] = CSH_NZ_END | (base & 0x7F800000u);
        }
        run += tot;
    }
}
__global__ void __launch_bounds__(128, 3) k_nzlist(EncCtx c) {
    // two waves per chunk, one half each: a wave loads its 128 blocks once, counts, and -- behind the barrier at which wave 0 reserves the
    // chunk's room -- writes its entries from the same registers
    CSH_SHARED uint32_t s_cnt[2];
    CSH_SHARED uint32_t s_rel;      // the chunk's first entry, relative to the list's region; 0xFFFFFFFF: no room
    CSH_WPERSIST(LV<uint4>, q, CSH_NZ_HALF, 2);
    const NzChunk ch = c.nzchunks[blockIdx.x];
    const int half = lwave();
    const bool skip = !(ch.levels & 1u) || (c.work_active && !c.work_active[ch.work0]);
    CSH_PHASE_LOOP(3) {
        if (skip) continue;
        const NzSet S = c.nzsets[ch.set];
        const uint32_t u0 = ch.j * 256u;
        const NzList L0 = c.nzlists[S.list[0]];
        if (phase == 0) {
            LV<uint32_t> cnt;
            LFOR(l) cnt[l] = 0u;
            nz_load_half(c, S, u0, half, q);
            nz_count_half(S, u0, half, q, cnt);
            const uint32_t n = lsum32(cnt);
            LFOR(l) if (l == 0) s_cnt[half] = n;
            continue;
        }
        if (phase == 1) {
            if (half == 0) {
                const uint32_t n0 = s_cnt[0] + s_cnt[1], n0a = (n0 + 3u) & ~3u, rec0 = L0.chunk0 + ch.j;
                uint32_t rel = 0;
                LFOR(l) if (l == 0) rel = atomicAdd(&c.nz_cursor[S.list[0]], n0a);
                rel = uni(rel);
                const bool ok0 = uint64_t(rel) + n0a <= L0.cap;
                LFOR(l) if (l == 0) { c.nz_chunk_off[rec0] = rel; c.nz_chunk_cnt[rec0] = ok0 ? n0 : 0u; if (!ok0) c.overflow[1] = 1; s_rel = ok0 ? rel : 0xFFFFFFFFu; }
            }
            continue;
        }
        if (s_rel == 0xFFFFFFFFu) continue;
        uint32_t *dst = c.nz_pool + L0.base + s_rel;
        uint8_t *blk_cnt = (c.nz_blk_cnt && S.cnt_base != 0xFFFFFFFFu) ? c.nz_blk_cnt + S.cnt_base : nullptr;
        uint16_t *blk_off = (blk_cnt && c.nz_blk_off) ? c.nz_blk_off + S.cnt_base : nullptr;
        nz_write_half(S, u0, half, q, dst, half ? s_cnt[0] : 0u, blk_cnt, blk_off);
        if (half == 1) {   // padding to the next 16-byte boundary: entries that code nothing
            const uint32_t n0 = s_cnt[0] + s_cnt[1], n0a = (n0 + 3u) & ~3u;
            LFOR(l) if (n0 + uint32_t(l) < n0a) dst[n0 + uint32_t(l)] = 0u;
        }
    }
}
// the other point transforms of a chunk, filtered from its level 0 (which this stage's k_nzlist or an earlier stage's made): count, one
// atomic add per list, write.  Four entries per lane and step; the second pass reads the chunk out of the L2.
__global__ void __launch_bounds__(64) k_nzfilter(EncCtx c) {
    const NzChunk ch = c.nzchunks[blockIdx.x];
    const uint32_t others = ch.levels & ~1u;   // levels 1..3 to filter, and CSH_NZ_COMPACT0
    if (!others) return;
    const bool compact0 = (ch.levels & CSH_NZ_COMPACT0) != 0u;
    const NzSet S = c.nzsets[ch.set];
    if (c.work_active && !c.work_active[ch.work0]) return;
    const NzList L0 = c.nzlists[S.list[0]];
    const uint32_t n0 = c.nz_chunk_cnt[L0.chunk0 + ch.j];
    const uint32_t *src = c.nz_pool + L0.base + c.nz_chunk_off[L0.chunk0 + ch.j];
    auto kept = [](uint32_t x, int L) { return ((x & CSH_NZ_END) != 0u || (((x >> 8) & 0x7FFFu) >> L) != 0u) ? 1u : 0u; };   // an END entry, or a coefficient that is not zero at level L
    // (level 0 here only under CSH_NZ_COMPACT0: the list k_trellis_ac wrote its levels into -- the coefficients it dropped have magnitude 0 -- is
    // compacted IN PLACE: a step's 256 entries are in registers before any of them is written, and an entry never moves up)
    const int Lfirst = compact0 ? 0 : 1;
    LV<uint32_t> cntL[CSH_NZ_LEVELS];
    LFOR(l) for (int L = 0; L < CSH_NZ_LEVELS; L++) cntL[L][l] = 0u;
    for (uint32_t g0 = 0; g0 < n0; g0 += 256) {
        LFOR(l) {
            const uint32_t g = g0 + 4u * uint32_t(l);
            uint4 e; e.x = e.y = e.z = e.w = 0u;
            if (g < n0) e = *reinterpret_cast<const uint4 *>(src + g);
            if (g < n0) {   // (the padding behind the chunk's last entry is not an entry: it must not count as a level-0 END)
                CSH_UNROLL
                for (int L = 0; L < CSH_NZ_LEVELS; L++) cntL[L][l] += kept(e.x, L) + kept(e.y, L) + kept(e.z, L) + kept(e.w, L);
            }
        }
    }
    uint32_t *dstL[CSH_NZ_LEVELS];
    uint32_t runL[CSH_NZ_LEVELS];
    bool okL[CSH_NZ_LEVELS];
    uint32_t n0new = n0;
    dstL[0] = nullptr; runL[0] = 0; okL[0] = false;
    if (compact0 && n0) {   // same place, same room: nothing to reserve; the count and the padding are written behind the last step
        dstL[0] = c.nz_pool + L0.base + c.nz_chunk_off[L0.chunk0 + ch.j];
        okL[0] = true;
        n0new = lsum32(cntL[0]);
    }
    for (int L = 1; L < CSH_NZ_LEVELS; L++) {
        dstL[L] = nullptr; runL[L] = 0; okL[L] = false;
        if (!((others >> L) & 1u)) continue;
        const NzList LL = c.nzlists[S.list[L]];
        const uint32_t n = lsum32(cntL[L]), na = (n + 3u) & ~3u;
        uint32_t rel = 0;
        LFOR(l) if (l == 0) rel = atomicAdd(&c.nz_cursor[S.list[L]], na);
        rel = uni(rel);
        okL[L] = n0 != 0u && uint64_t(rel) + na <= LL.cap;
        dstL[L] = c.nz_pool + LL.base + rel;
        LFOR(l) {
            if (l == 0) { c.nz_chunk_off[LL.chunk0 + ch.j] = rel; c.nz_chunk_cnt[LL.chunk0 + ch.j] = okL[L] ? n : 0u; if (!okL[L]) c.overflow[1] = 1; }
            if (okL[L] && n + uint32_t(l) < na) dstL[L][n + uint32_t(l)] = 0u;   // padding
        }
    }
    for (uint32_t g0 = 0; g0 < n0; g0 += 256) {
        LV<uint32_t> e0, e1, e2, e3;
        LFOR(l) {
            const uint32_t g = g0 + 4u * uint32_t(l);
            uint4 e; e.x = e.y = e.z = e.w = 0u;
            if (g < n0) e = *reinterpret_cast<const uint4 *>(src + g);
            e0[l] = e.x; e1[l] = e.y; e2[l] = e.z; e3[l] = e.w;
        }
        for (int L = Lfirst; L < CSH_NZ_LEVELS; L++) {
            if (!okL[L]) continue;
            LV<uint32_t> k4;
            LFOR(l) k4[l] = kept(e0[l], L) + kept(e1[l], L) + kept(e2[l], L) + kept(e3[l], L);
            uint32_t tot;
            const LV<uint32_t> ex = lscan(k4, tot);
            LFOR(l) {
                const uint32_t e[4] = {e0[l], e1[l], e2[l], e3[l]};
                uint32_t o = runL[L] + ex[l];
                CSH_UNROLL
                for (int q = 0; q < 4; q++)
                    if (kept(e[q], L)) dstL[L][o++] = (e[q] & 0xFF8000FFu) | ((((e[q] >> 8) & 0x7FFFu) >> L) << 8);
            }
            runL[L] += tot;
        }
    }
    if (okL[0]) {
        const uint32_t na = (n0new + 3u) & ~3u;
        LFOR(l) {
            if (n0new + uint32_t(l) < na) dstL[0][n0new + uint32_t(l)] = 0u;   // padding to the 16-byte boundary: entries that code nothing
            if (l == 0) c.nz_chunk_cnt[L0.chunk0 + ch.j] = n0new;
        }
    }
}

// ------------------------------------------------------------------------------------------------ the events of a first-pass scan
// What entry e means in scan (Ss, Se), given the entry p in front of it (the END of the block before: CSH_NZ_END):
//   coded   Ss <= k <= Se: symbol (run & 15) << 4 | size, run >> 4 ZRLs in front of it, `size` value bits
//   term    the block's first entry behind the band (the END entry included): the block's end -- has: it coded something;
//           eob: it ends with an EOB (nothing coded at Se)
struct NzEvent { uint32_t coded, term, run, size, bits, has, eob, blk; };
__device__ __forceinline__ static NzEvent nz_event(uint32_t e, uint32_t p, uint32_t Ss, uint32_t Se) {
    NzEvent v;
    const uint32_t k = e & 127u, pk = p & 127u;
    const bool same = !(p & CSH_NZ_END);            // p belongs to e's block (every block's last entry is its END)
    const bool pin = same && pk >= Ss;               // ... and to the band (pk < k: the list is sorted)
    v.coded = (k >= Ss && k <= Se) ? 1u : 0u;
    v.term = (k > Se && !(same && pk > Se)) ? 1u : 0u;
    v.run = k - (pin ? pk + 1u : Ss);
    const uint32_t m = (e >> 8) & 0x7FFFu;
    v.size = lbitlen(m);
    v.bits = ((e & 128u) ? ~m : m) & ((1u << v.size) - 1u);
    v.has = pin ? 1u : 0u;                           // (term: pk <= Se)
    v.eob = (pin && pk == Se) ? 0u : 1u;
    v.blk = (e >> 23) & 255u;
    return v;
}
struct ListSlot { const uint32_t *lst; uint32_t n; };
__device__ __forceinline__ static ListSlot list_of_slot(const EncCtx &c, const SlotRec &r) {
    // the slot record names its list and its chunk record (k_make_slots): three independent loads here, not a chain of four (work item -> list
    // -> chunk record -> offset); the waves of these kernels are short, and what they wait for first is this
    ListSlot s;
    s.n = c.nz_chunk_cnt[r.nzrec];
    s.lst = c.nz_pool + c.nzlists[r.nzlist].base + c.nz_chunk_off[r.nzrec];
    return s;
}
// four consecutive entries per lane (the chunk starts on a 16-byte boundary and is padded to one with entries that code nothing)
__device__ __forceinline__ static void list_load4(const ListSlot &s, uint32_t g, uint32_t &e0, uint32_t &e1, uint32_t &e2, uint32_t &e3) {
    uint4 q; q.x = q.y = q.z = q.w = 0u;
    if (g < s.n) q = *reinterpret_cast<const uint4 *>(s.lst + g);
    e0 = q.x; e1 = q.y; e2 = q.z; e3 = q.w;
}

// ---- statistics: ONE WAVE per (scan, chunk) slot.  Symbol histogram in LDS (four copies, lane & 3: the frequent symbols -- 0x01, 0x11,
// 0x02 -- meet in most steps, and equal addresses serialise), raw-bit count, the has-symbol / ends-with-EOB bits of the chunk's 256 blocks.
__global__ void __launch_bounds__(256) k_list_stats(EncCtx c) {
    CSH_SHARED uint32_t s_hist[4][4][256];
    CSH_SHARED uint32_t s_flag[4][16];    // words 0..7: has-symbol, 8..15: ends-with-EOB
    const int wv = lwave();
    const uint32_t idx = blockIdx.x * 4u + uint32_t(wv);
    if (idx >= c.nlist_slots) return;
    const uint32_t cs = c.list_slots[idx];
    const SlotRec r = c.slots[cs];
    if (c.work_active && !c.work_active[r.work]) return;
    const ListSlot ls = list_of_slot(c, r);
    uint32_t *hist = &s_hist[wv][0][0], *flag = s_flag[wv];
    LFOR(l) { for (int i = l; i < 1024; i += 64) hist[i] = 0u; if (l < 16) flag[l] = 0u; }
    CSP_WAVE_SYNC();
    const uint32_t Ss = r.Ss, Se = r.Se;
    uint32_t carry = CSH_NZ_END;
    LV<uint32_t> raw;
    LFOR(l) raw[l] = 0u;
    // (the next step's entries are asked for before this step's are looked at: a step is short, and what it waited for was its own load)
    LV<uint32_t> x0, x1, x2, x3;
    LFOR(l) list_load4(ls, 4u * uint32_t(l), x0[l], x1[l], x2[l], x3[l]);
    for (uint32_t g0 = 0; g0 < ls.n; g0 += 256) {
        const LV<uint32_t> e0 = x0, e1 = x1, e2 = x2, e3 = x3;
        if (g0 + 256u < ls.n) LFOR(l) list_load4(ls, g0 + 256u + 4u * uint32_t(l), x0[l], x1[l], x2[l], x3[l]);
        const LV<uint32_t> p0 = lprev(e3, carry);
        carry = llast(e3);
        LFOR(l) {
            uint32_t *h = hist + 256 * (l & 3);
            const uint32_t e[4] = {e0[l], e1[l], e2[l], e3[l]}, p[4] = {p0[l], e0[l], e1[l], e2[l]};
            CSH_UNROLL
            for (int q = 0; q < 4; q++) {
                const NzEvent v = nz_event(e[q], p[q], Ss, Se);
                if (v.coded) {
                    atomicAdd(&h[((v.run & 15u) << 4) | v.size], 1u);
                    if (v.run >> 4) atomicAdd(&h[0xF0], v.run >> 4);
                    raw[l] += v.size;
                } else if (v.term) {
                    if (v.has) atomicOr(&flag[v.blk >> 5], 1u << (v.blk & 31u));
                    if (v.eob) atomicOr(&flag[8u + (v.blk >> 5)], 1u << (v.blk & 31u));
                }
            }
        }
    }
    CSP_WAVE_SYNC();
    const uint32_t rawbits = lsum32(raw);
    LFOR(l) {
        for (int i = l; i < 256; i += 64) {
            const uint32_t v = hist[i] + hist[256 + i] + hist[512 + i] + hist[768 + i];
            c.slot_hist[size_t(r.hist_row) * 256u + uint32_t(i)] = uint16_t(v);
            if (v) atomicAdd(&c.tables[r.table_base].freq[i], v);
        }
        if (l == 0) c.slot_raw[cs] = rawbits;
        if (l < 4 && r.j * 4u + uint32_t(l) < ((r.nunits_work + 63u) >> 6)) {   // lane = block bit, so the chunk's flags ARE four words of the scan's bit vectors
            c.sym_bits[r.word_base + r.j * 4u + uint32_t(l)] = uint64_t(flag[2 * l]) | (uint64_t(flag[2 * l + 1]) << 32);
            c.eob_bits[r.word_base + r.j * 4u + uint32_t(l)] = uint64_t(flag[8 + 2 * l]) | (uint64_t(flag[9 + 2 * l]) << 32);
        }
    }
}

// ---- pack: ONE WAVE per (scan, chunk) slot.  256 entries a step, four per lane: their bits (ZRLs, symbol + value bits; a block's EOBRUN
// symbol at its end), a wave scan of the lengths, every lane ORs its pieces into the wave's LDS window of the bit stream.  The window is
// word-aligned with the raw pool, so flushing it is a plain copy: only the chunk's first and last word can be shared with a neighbouring
// chunk and are ORed in (k_zero_edges cleared them).
__device__ __forceinline__ static void lor_bits(uint32_t *words, uint64_t pos, uint32_t v, uint32_t n) {   // n in 1..32, at bit `pos` of a big-endian-logical word array
    v &= n >= 32 ? 0xFFFFFFFFu : ((1u << n) - 1u);
    const uint64_t t = uint64_t(v) << (64u - n - uint32_t(pos & 31u));
    const uint32_t hi = uint32_t(t >> 32), lo = uint32_t(t);
    if (hi) atomicOr(words + (pos >> 5), hi);
    if (lo) atomicOr(words + (pos >> 5) + 1, lo);
}
__global__ void __launch_bounds__(256) k_list_pack(EncCtx c) {
    CSH_SHARED uint32_t s_win[4][CSH_LP_WORDS];
    CSH_SHARED uint32_t s_lut[4][256];
    CSH_SHARED uint16_t s_eob[4][256];
    const int wv = lwave();
    const uint32_t idx = blockIdx.x * 4u + uint32_t(wv);
    if (idx >= c.nlist_slots) return;
    const uint32_t cs = c.list_slots[idx];
    const SlotRec r = c.slots[cs];
    if (c.work_active && !c.work_active[r.work]) return;
    const ScanWork &w = c.work[r.work];
    if (w.no_room) { LFOR(l) if (l == 0) c.status[w.image] = 20200; return; }   // decided per scan by k_scan_place
    const ListSlot ls = list_of_slot(c, r);
    // the chunk's place: bits [raw_bit0, raw_bit0 + nbits) of the raw pool; the scan's last chunk also carries the 1-bits that fill the last byte
    const uint64_t scan0 = c.chunk_off[r.first_chunk];
    const uint64_t raw_bit0 = w.raw_off * 8 + (c.chunk_off[cs] - scan0);
    uint32_t pad = 0;
    if (r.j == r.nch - 1) { const uint64_t total = c.chunk_off[r.first_chunk + r.nch] - scan0; pad = uint32_t((8 - (total & 7)) & 7); }
    uint32_t *buf = s_win[wv], *lut = s_lut[wv];
    uint16_t *eobrun = s_eob[wv];
    LFOR(l) {
        for (int i = l; i < 256; i += 64) {
            lut[i] = c.tables[r.table_base].lut[i];
            eobrun[i] = uint32_t(i) < r.nun ? c.eobrun[r.unit0 + uint32_t(i)] : uint16_t(0);
        }
        for (int i = l; i < CSH_LP_WORDS; i += 64) buf[i] = 0u;
    }
    CSP_WAVE_SYNC();
    uint32_t *out = c.raw + (raw_bit0 >> 5);        // word 0 of the frame below
    uint64_t pos = raw_bit0 & 31u;                   // next bit, in the frame whose word 0 is the chunk's first word in the pool
    uint32_t ww = 0;                                 // first word of the window
    bool first_flush = true;
    const uint32_t Ss = r.Ss, Se = r.Se;
    const uint32_t zrl = lut[0xF0], zc = zrl & 0xFFFFu, zl = zrl >> 16;
    uint32_t carry = CSH_NZ_END;
    const uint32_t n_ext = ls.n + (pad ? 1u : 0u);   // the byte fill rides as one more entry
    LV<uint32_t> x0, x1, x2, x3;   // the next step's entries, asked for a step ahead
    LFOR(l) list_load4(ls, 4u * uint32_t(l), x0[l], x1[l], x2[l], x3[l]);
    for (uint32_t g0 = 0; g0 < n_ext; g0 += 256) {
        const LV<uint32_t> e0 = x0, e1 = x1, e2 = x2, e3 = x3;
        if (g0 + 256u < n_ext) LFOR(l) list_load4(ls, g0 + 256u + 4u * uint32_t(l), x0[l], x1[l], x2[l], x3[l]);
        const LV<uint32_t> p0 = lprev(e3, carry);
        carry = llast(e3);
        // what every entry emits: nz[q] ZRLs, then the n[q] low bits of v[q]
        LV<uint32_t> v0, v1, v2, v3, n0, n1, n2, n3, z, len;
        LFOR(l) {
            const uint32_t e[4] = {e0[l], e1[l], e2[l], e3[l]}, p[4] = {p0[l], e0[l], e1[l], e2[l]};
            uint32_t v[4], n[4], zz = 0, ln = 0;
            CSH_UNROLL
            for (int q = 0; q < 4; q++) {
                v[q] = 0; n[q] = 0;
                const uint32_t g = g0 + 4u * uint32_t(l) + uint32_t(q);
                const NzEvent ev = nz_event(e[q], p[q], Ss, Se);
                if (g >= ls.n) { if (pad && g == ls.n) { v[q] = (1u << pad) - 1u; n[q] = pad; } }
                else if (ev.coded) {
                    const uint32_t t = lut[((ev.run & 15u) << 4) | ev.size];
                    v[q] = ((t & 0xFFFFu) << ev.size) | ev.bits; n[q] = (t >> 16) + ev.size;   // <= 16 + 15 bits
                    zz |= (ev.run >> 4) << (2 * q);
                    ln += (ev.run >> 4) * zl;
                } else if (ev.term) {
                    const uint32_t run = eobrun[ev.blk];
                    if (run) {
                        const uint32_t nb = lbitlen(run) - 1u, t = lut[nb << 4];
                        v[q] = ((t & 0xFFFFu) << nb) | (run & ((1u << nb) - 1u)); n[q] = (t >> 16) + nb;   // <= 16 + 14 bits
                    }
                }
                ln += n[q];
            }
            v0[l] = v[0]; v1[l] = v[1]; v2[l] = v[2]; v3[l] = v[3]; n0[l] = n[0]; n1[l] = n[1]; n2[l] = n[2]; n3[l] = n[3]; z[l] = zz; len[l] = ln;
        }
        uint32_t tot;
        const LV<uint32_t> ex = lscan(len, tot);
        LFOR(l) {
            uint64_t at = pos + ex[l] - uint64_t(ww) * 32u;    // bit position inside the window
            const uint32_t v[4] = {v0[l], v1[l], v2[l], v3[l]}, n[4] = {n0[l], n1[l], n2[l], n3[l]};
            CSH_UNROLL
            for (int q = 0; q < 4; q++) {
                for (uint32_t t = (z[l] >> (2 * q)) & 3u; t; t--) { lor_bits(buf, at, zc, zl); at += zl; }
                if (n[q]) { lor_bits(buf, at, v[q], n[q]); at += n[q]; }
            }
        }
        pos += tot;
        // slide the window when another step's worth of bits might not fit any more
        const uint32_t done = uint32_t(pos >> 5) - ww;   // complete words in the window
        if (done > CSH_LP_WORDS - 640) {
            CSP_WAVE_SYNC();
            LFOR(l) for (uint32_t q = uint32_t(l); q < done; q += 64) {
                const uint32_t v = buf[q];
                if (first_flush && q == 0) { if (v) atomicOr(out + ww, v); } else out[ww + q] = v;
            }
            const uint32_t partial = buf[done];
            CSP_WAVE_SYNC();
            LFOR(l) for (int q = l; q < CSH_LP_WORDS; q += 64) buf[q] = (q == 0) ? partial : 0u;
            CSP_WAVE_SYNC();
            ww += done; first_flush = false;
        }
    }
    CSP_WAVE_SYNC();
    const uint32_t last = uint32_t((pos + 31) >> 5) - ww;   // words in the window that carry bits
    LFOR(l) for (uint32_t q = uint32_t(l); q < last; q += 64) {
        const uint32_t v = buf[q];
        if ((first_flush && q == 0) || q == last - 1) { if (v) atomicOr(out + ww + q, v); } else out[ww + q] = v;
    }
}

__global__ void k_reset_works(ScanWork *work, int nwork) {
    const int j = int(blockIdx.x * blockDim.x + threadIdx.x);
    if (j >= nwork) return;
    work[j].out_off = 0xFFFFFFFFu; work[j].raw_bytes = 0; work[j].hdr_bytes = 0; work[j].ff_bytes = 0; work[j].no_room = 0;
}

void launch_nzlist(hipStream_t st, const EncCtx &c) {
    if (!c.nnzchunks) return;
    CSH_LAUNCH_PHASED(k_nzlist, 3, dim3(c.nnzchunks), dim3(2 * CSP_WAVE_THREADS), st, c);
    CSH_LAUNCH(k_nzfilter, dim3(c.nnzchunks), dim3(CSP_WAVE_THREADS), st, c);
}
void launch_list_stats(hipStream_t st, const EncCtx &c) { if (c.nlist_slots) CSH_LAUNCH(k_list_stats, dim3((c.nlist_slots + 3) / 4), dim3(4 * CSP_WAVE_THREADS), st, c); }
void launch_list_pack(hipStream_t st, const EncCtx &c) { if (c.nlist_slots) CSH_LAUNCH(k_list_pack, dim3((c.nlist_slots + 3) / 4), dim3(4 * CSP_WAVE_THREADS), st, c); }
// ---- the slots of the work items: one workgroup per work item, one lane per 256-unit chunk.  Under the scan search a 1080p image has ~3.9 k slots in 58 work
// items: built on the host they were 64 MB of records per 256 files to write and to upload in front of the first kernel (the boundary call paid ~15 ms of
// every 50 for them); the host only counts them now (pipeline.cpp add_works).  Slots between the stages belong to no work item and stay zero.
__global__ void __launch_bounds__(64) k_make_slots(const ScanWork *works, uint32_t nworks, const EncScan *script, const NzList *nzlists, SlotRec *slots, uint32_t *slot_work, uint32_t *list_slots,
                                                   uint32_t *tok_slots) {
    const uint32_t wi = blockIdx.x;
    if (wi >= nworks) return;
    const ScanWork &w = works[wi];
    const EncScan &e = script[w.scan];
    const uint32_t nch = (w.nunits + 255u) / 256u;
    const bool prog_ac = e.Ss > 0 && !e.sequential, listed = w.list != 0xFFFFFFFFu;
    for (uint32_t j = threadIdx.x; j < nch; j += blockDim.x) {
        SlotRec r;
        r.work = wi; r.j = j; r.nch = nch; r.first_chunk = w.first_chunk; r.unit0 = w.unit_base + 256u * j;
        r.nun = w.nunits - 256u * j < 256u ? w.nunits - 256u * j : 256u; r.table_base = w.table_base; r.ntables = uint16_t(e.ntables);
        r.flags = uint16_t((prog_ac ? 1 : 0) | (prog_ac && e.Ah ? 2 : 0) | (listed ? 4 : 0));
        r.hist_row = w.hist_row0 + j * uint32_t(e.ntables);
        r.word_base = w.word_base; r.unit_base = w.unit_base; r.nunits_work = w.nunits;
        r.Ss = uint8_t(e.Ss); r.Se = uint8_t(e.Se); r.Ah = uint8_t(e.Ah); r.Al = uint8_t(e.Al); r.corr0 = w.corr_base == 0xFFFFFFFFu ? 0u : w.corr_base + 256u * j;
        r.nzlist = listed ? w.list : 0u; r.nzrec = listed ? nzlists[w.list].chunk0 + j : 0u;
        slots[w.first_chunk + j] = r;
        slot_work[w.first_chunk + j] = wi;
        (listed ? list_slots : tok_slots)[w.ls_base + j] = w.first_chunk + j;
    }
}
// the scan search re-points work items to the lists of the point transform it chose (pipeline.cpp search_decide): their slots follow
__global__ void __launch_bounds__(64) k_rebind_slots(const ScanWork *works, uint32_t nworks, const NzList *nzlists, SlotRec *slots) {
    const uint32_t wi = blockIdx.x;
    if (wi >= nworks) return;
    const ScanWork &w = works[wi];
    if (w.list == 0xFFFFFFFFu) return;
    const uint32_t nch = (w.nunits + 255u) / 256u, chunk0 = nzlists[w.list].chunk0;
    for (uint32_t j = threadIdx.x; j < nch; j += blockDim.x) { slots[w.first_chunk + j].nzlist = w.list; slots[w.first_chunk + j].nzrec = chunk0 + j; }
}
void launch_rebind_slots(hipStream_t st, const ScanWork *works, uint32_t nworks, const NzList *nzlists, SlotRec *slots) {
    if (nworks) CSH_LAUNCH(k_rebind_slots, dim3(nworks), dim3(64), st, works, nworks, nzlists, slots);
}
void launch_make_slots(hipStream_t st, const ScanWork *works, uint32_t nworks, const EncScan *script, const NzList *nzlists, SlotRec *slots, uint32_t *slot_work, uint32_t *list_slots,
                       uint32_t *tok_slots) {
    if (!nworks) return;
    CSH_LAUNCH(k_make_slots, dim3(nworks), dim3(64), st, works, nworks, script, nzlists, slots, slot_work, list_slots, tok_slots);
}

void launch_reset_works(hipStream_t st, ScanWork *work, int nwork) { if (nwork) CSH_LAUNCH(k_reset_works, dim3((nwork + 255) / 256), dim3(256), st, work, nwork); }

}  // namespace csh
