// Attention kernels for gfx950.
//
// flash_kernel: softmax(Q K^T * scale [+ causal mask]) V over packed variable-length sequences
// (reference: dense masked attention at AudioEncoder.swift:83-91 with the block mask of :303-320,
// and QuantizedTextDecoder.swift:188-201 with the causal mask of :364-381 and GQA tiling :171-185;
// the additive -1e9 masks are realised by skipping the masked keys, which is what they evaluate
// to in fp32).  One workgroup = 64 queries x one head, 4 waves x 16 queries; K/V tiles of 32 keys
// are register-staged into LDS (K row-major, V transposed).  The scores are computed transposed
// (S^T = K Q^T) so each lane owns one query column: the row statistics need two shuffles, and the
// exponentiated accumulator is directly the B operand of O^T = V^T P^T (no LDS round trip for P).
//
// decode_attn_kernel: one new query per utterance against its KV cache (GQA group shares the
// K/V stream).  HBM-bound: K and V are requested as whole 256-byte rows (a load instruction that takes a
// 64-byte piece of 16 rows -- the MFMA A-fragment shape -- touches every 128-byte line twice, and a compute unit
// then sustains 34 instead of 50 GB/s: measured with the bare request patterns, scripts/probe_cu_mask.hip); the
// K rows become A fragments of S^T = K Q^T through a wave-private LDS tile; the exponentiated scores, rounded to bf16,
// are as they stand the B operand of O^T = V^T P^T, whose A fragments are transposing LDS reads of the V tile.  The
// vector ALU only does the softmax: with the products on it (fp32 FMAs over broadcast probabilities) a compute
// unit could not issue fast enough to keep its share of the HBM stream when the kernel is confined to 160 CUs.
#include "attn.h"
#include <cstdlib>
#include "prof.h"
#include "rope.h"
#include <algorithm>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;

namespace {

union Frag {
    bf16x8 v;
    uint4 u;
    uint2 h[2];
};

// Reductions over the four 16-lane rows of a wave (lanes l, l^16, l^32, l^48), result in every lane: one
// v_permlane16_swap for the row pairs and one v_permlane32_swap for the wave halves -- two vector-ALU moves, no trip
// through the LDS crossbar and no address arithmetic.  swap(x, x) returns (rows 0 0 2 2, rows 1 1 3 3) resp.
// (lower half twice, upper half twice).
__device__ __forceinline__ float rows_max(float x) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    x = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));                                 // lanes l ^ 16
    const auto h = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return fmaxf(__uint_as_float(h[0]), __uint_as_float(h[1]));                              // lanes l ^ 32
}
__device__ __forceinline__ float rows_sum(float x) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    x = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    const auto h = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(h[0]) + __uint_as_float(h[1]);
}

// G: query heads of one KV head that a workgroup computes against each staged K/V tile (1 or 2)
template <int D, bool CAUSAL, int G>
__global__ __launch_bounds__(256, G == 2 ? 3 : 1) void flash_kernel(FlashParams p) {
    constexpr int KS = D * 2 + 16;  // K row stride in LDS (bytes)
    constexpr int VS = D * 2 + 32;  // V row stride: the transposing reads of a 32-lane half (8 rows x 4 column quads) then hit 64 distinct banks
    constexpr int KCH = D / 8;      // 16-byte chunks per row
    constexpr int NLD = 32 * KCH / 256;
    constexpr int NKS = D / 32, NDT = D / 16;
    __shared__ __attribute__((aligned(16))) char lds[32 * KS + 32 * VS];
    char* kl = lds;
    char* vl = lds + 32 * KS;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lg = lane >> 4;
    const AttnTile tile = p.tiles[blockIdx.x];
    const AttnSeq seq = p.seqs[tile.seq];
    const int h0 = blockIdx.y * G, kvh = h0 / p.group;
    const int len = seq.len, q0 = tile.q0;
    const int qi = q0 + 16 * wave + li;
    const int64_t qrow = seq.q_row0 + min(qi, len - 1);

    Frag qf[G][NKS];
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) qf[g][ks].u = *(const uint4*)(p.Q + qrow * p.q_ld + (h0 + g) * D + 32 * ks + 8 * lg);

    const int kv_end = CAUSAL ? min(len, q0 + 64) : len;
    const int ntiles = (kv_end + 31) >> 5;
    const bf16_t* Kb = p.K + seq.kv_off + (int64_t)kvh * p.kv_hs;
    const bf16_t* Vb = p.V + seq.kv_off + (int64_t)kvh * p.kv_hs;

    uint4 kreg0, kreg1, vreg0, vreg1;
    const int k_key0 = tid / KCH, k_ch0 = tid % KCH, k_key1 = (tid + 256) / KCH, k_ch1 = (tid + 256) % KCH;
#define FLASH_LOAD(t)                                                                                   \
    {                                                                                                   \
        const int kv0_ = (t) * 32;                                                                      \
        const int64_t r0_ = (int64_t)min(kv0_ + k_key0, len - 1) * p.kv_ld + k_ch0 * 8;                 \
        kreg0 = *(const uint4*)(Kb + r0_);                                                              \
        vreg0 = *(const uint4*)(Vb + r0_);                                                              \
        if constexpr (NLD > 1) {                                                                        \
            const int64_t r1_ = (int64_t)min(kv0_ + k_key1, len - 1) * p.kv_ld + k_ch1 * 8;             \
            kreg1 = *(const uint4*)(Kb + r1_);                                                          \
            vreg1 = *(const uint4*)(Vb + r1_);                                                          \
        }                                                                                               \
    }
    // V stays row-major in LDS; the A fragments of O^T = V^T P^T are transposing reads (per 16 lanes a block of 4 keys x 16
    // dims, lane 4q + p supplying the address of key q, dims 4p..4p+3, lane i receiving dim i of the 4 keys)
    const char* vr = vl + (4 * lg + (li >> 2)) * VS + (li & 3) * 8;
    FLASH_LOAD(0);

    const float sl2 = p.scale * 1.44269504088896340736f;
    float m[G], l[G];
    f32x4 o[G][NDT];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        m[g] = -INFINITY;
        l[g] = 0.f;
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) o[g][dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }

    for (int t = 0; t < ntiles; ++t) {
        __syncthreads();
        *(uint4*)(kl + k_key0 * KS + k_ch0 * 16) = kreg0;
        if constexpr (NLD > 1) *(uint4*)(kl + k_key1 * KS + k_ch1 * 16) = kreg1;
        *(uint4*)(vl + k_key0 * VS + k_ch0 * 16) = vreg0;
        if constexpr (NLD > 1) *(uint4*)(vl + k_key1 * VS + k_ch1 * 16) = vreg1;
        __syncthreads();
        if (t + 1 < ntiles) FLASH_LOAD(t + 1);
        const int kv0 = t * 32;
        if (CAUSAL && kv0 > q0 + 16 * wave + 15) continue;  // wave-uniform: every key is in the future

        // every K and V fragment is read from LDS once and used for all G heads
        f32x4 s[G][2];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
#pragma unroll
            for (int g = 0; g < G; ++g) s[g][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                Frag kf;
                kf.u = *(const uint4*)(kl + (16 * nt + li) * KS + (lg + 4 * ks) * 16);
#pragma unroll
                for (int g = 0; g < G; ++g) s[g][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf.v, qf[g][ks].v, s[g][nt], 0, 0, 0);
            }
        }
        Frag pf[G];
        float alpha[G];
#pragma unroll
        for (int g = 0; g < G; ++g) {
            float sv[8], tmax = -INFINITY;
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = kv0 + 16 * nt + 4 * lg + r;
                    const bool ok = key < len && (!CAUSAL || key <= qi);
                    sv[nt * 4 + r] = ok ? s[g][nt][r] * sl2 : -INFINITY;
                    tmax = fmaxf(tmax, sv[nt * 4 + r]);
                }
            tmax = rows_max(tmax);
            const float m_new = fmaxf(m[g], tmax);
            alpha[g] = exp2f(m[g] - m_new);
            float ps = 0.f, pe[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                pe[j] = exp2f(sv[j] - m_new);
                ps += pe[j];
            }
            pf[g].u = make_uint4(pack_bf16x2(pe[0], pe[1]), pack_bf16x2(pe[2], pe[3]), pack_bf16x2(pe[4], pe[5]),
                                 pack_bf16x2(pe[6], pe[7]));
            l[g] = l[g] * alpha[g] + ps;
            m[g] = m_new;
        }
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) {
            union { s16x4 v; uint2 u; } lo, hi;   // keys 4 lg .. 4 lg + 3 and 16 + 4 lg .. of dim 16 dt + li
            lo.v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(vr + 32 * dt));
            hi.v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(vr + 16 * VS + 32 * dt));
            Frag vf;
            vf.h[0] = lo.u;
            vf.h[1] = hi.u;
#pragma unroll
            for (int g = 0; g < G; ++g) {
                o[g][dt] *= alpha[g];
                o[g][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf.v, pf[g].v, o[g][dt], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int g = 0; g < G; ++g) {
        float lt = l[g];
        lt = rows_sum(lt);
        if (qi < len) {
            const float inv = 1.0f / lt;
            bf16_t* orow = p.O + (seq.q_row0 + qi) * p.o_ld + (h0 + g) * D + 4 * lg;
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt)
                *(uint2*)(orow + 16 * dt) = make_uint2(pack_bf16x2(o[g][dt][0] * inv, o[g][dt][1] * inv),
                                                       pack_bf16x2(o[g][dt][2] * inv, o[g][dt][3] * inv));
        }
    }
}

#undef FLASH_LOAD

// WAVES = 4 for large groups (three workgroups share a CU); 8 when there are fewer (utterance, head) items than CUs,
// so that one item's keys are streamed by twice as many waves.
template <int GROUP, bool FUSED, int WAVES>
__global__ __launch_bounds__(64 * WAVES, WAVES == 4 ? 3 : 1) void decode_attn_kernel(DecodeAttnParams p) {
    constexpr int D = 128;
    __shared__ float sm_m[WAVES][GROUP], sm_l[WAVES][GROUP];
    __shared__ __attribute__((aligned(16))) float so[WAVES][GROUP][D];
    // this step's queries (FUSED: normalised and rotated here), as the 16 rows of the B operand: rows >= GROUP stay zero;
    // read per tile rather than held in registers (the accumulators and two tiles in flight need them)
    __shared__ __attribute__((aligned(16))) bf16_t sq[16][D + 8];
    __shared__ __attribute__((aligned(16))) float snew[2][D];     // FUSED: the new key (rotated) and value, bf16-rounded
    // per wave: one 16-key K tile, rows padded to 272 bytes so that both the row-wise writes and the fragment reads
    // (16 rows, one 16-byte chunk each) spread over all banks
    constexpr int KROW = 2 * D + 16;
    __shared__ __attribute__((aligned(16))) char ktile[WAVES][16 * KROW];
    // and one V tile, rows padded to 288 bytes: the transposing read takes, per 16 lanes, 4 rows x 16 columns with
    // four lanes (32 bytes) per row, and rows 32 bytes apart modulo the banks keep the two groups of a half-wave apart
    constexpr int VROW = 2 * D + 32;
    __shared__ __attribute__((aligned(16))) char vtile[WAVES][16 * VROW];
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, lg = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // uniform: tile indices and loop branches stay scalar
    char* const kw = ktile[wave] + (4 * lg) * KROW + 16 * li;        // this lane's chunk of rows 4 lg .. 4 lg + 3
    const char* const kr = ktile[wave] + li * KROW + 16 * lg;        // fragment ks of key row li: + 64 ks
    const char* const qr = (const char*)&sq[li][8 * lg];             // fragment ks of query li: + 64 ks
    for (int idx = tid; idx < 16 * (D + 8) / 2; idx += 64 * WAVES) ((uint32_t*)&sq[0][0])[idx] = 0u;
    __syncthreads();
    char* const vw = vtile[wave] + (4 * lg) * VROW + 16 * li;
    // lane 4 q + p of a 16-lane group addresses row q, columns 4 p .. 4 p + 3 of the group's block (keys 4 lg .. 4 lg + 3,
    // dims 16 dt .. 16 dt + 15) and receives column (lane & 15) of the four rows: the A fragment of a 16x16x16 MFMA
    const char* const vr = vtile[wave] + (4 * lg + (li >> 2)) * VROW + 8 * (li & 3);   // + 32 dt
    // (utterance, kv head) work items, strided over the grid: the launcher may cap the grid so that a fixed
    // number of workgroups per CU stream the cache while another stream's kernels share the CUs
    const int items = p.B * p.kv_heads;
    for (int item = blockIdx.x; item < items; item += gridDim.x) {
    const int row = item % p.B, kvh = item / p.B;
    const int u = p.slot ? p.slot[row] : row;
    if (p.done && p.done[u]) {   // workgroup-uniform: a stopped utterance costs no cache traffic
        for (int idx = tid; idx < GROUP * D / 2; idx += 64 * WAVES)
            *(uint32_t*)(p.O + (int64_t)row * p.o_ld + (kvh * GROUP) * D + 2 * idx) = 0u;
        continue;
    }
    // FUSED: len counts the keys already cached; the new one is handled from LDS after the loop
    const int len = p.lens[u] + (FUSED ? 0 : p.len_add);
    const bf16_t* Kb = p.K + (int64_t)u * p.utt_stride + (int64_t)kvh * p.head_stride;
    const bf16_t* Vb = p.V + (int64_t)u * p.utt_stride + (int64_t)kvh * p.head_stride;
    const float sl2 = p.scale * 1.44269504088896340736f;
    float m = -INFINITY, l = 0.f;
    f32x4 o[D / 16];   // O^T accumulators: lane (q, lg) holds dims 16 dt + 4 lg + r of query q (lanes q >= GROUP are padding)
#pragma unroll
    for (int dt = 0; dt < D / 16; ++dt) o[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // Tiles of 16 keys, round-robin over the waves.  A tile's K rows wait in registers until the previous tile's scores
    // are done, move to LDS, and the registers are at once reloaded with the next tile's K rows; the V rows follow the
    // same way after the softmax.  So about one tile (8 KB) per wave is in flight at any time behind one register copy.
    // (A second register set that requests the K rows two tiles ahead was measured at 2048 clips: with three workgroups
    // per CU the 168-register budget spills and a launch takes 592 us instead of 460; with two workgroups per CU and no
    // spill 485 us; K and V both two tiles ahead, 207 registers, two workgroups per CU: 482 us (128 clips of the 0.6B
    // model: 60.2 against 61.5 us).  Twelve waves per CU with one tile each draw more than eight with two.  Sixteen
    // waves -- four workgroups per CU, LDS cut to 37 KB by merging through the K tiles and keeping one zero row for the
    // padding queries -- need 128 registers, which spills 16: 539 us.)
    // Loads use one base address per tile plus immediate offsets (no per-row clamping) and are marked non-temporal:
    // the cache is read once per step, and streaming loads that do not displace L2 lines sustain ~15 % more HBM
    // bandwidth on this GPU than plain ones (6.9 against 6.0 TB/s in a bare read loop).
    // Rows past len-1 are allocated padding (the cache holds 16 spare rows); they are masked in the last tile.
#define DA_LOAD_K(T)                                                                                           \
    {                                                                                                          \
        const bf16_t* kp_ = Kb + (int64_t)((T) * 16 + 4 * lg) * D + 8 * li;   /* 16 lanes per row */         \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                          \
            kf[i].v = __builtin_nontemporal_load((const bf16x8*)(kp_ + i * D));                               \
    }
#define DA_LOAD_V(T)                                                                                           \
    {                                                                                                          \
        const bf16_t* vp_ = Vb + (int64_t)((T) * 16 + 4 * lg) * D + 8 * li;                                    \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                          \
            vf[i].v = __builtin_nontemporal_load((const bf16x8*)(vp_ + i * D));                               \
    }
    // LAST: the wave's last tile (nothing more to request; keys past len are masked)
#define DA_TILE(T, LAST)                                                                                       \
    {                                                                                                          \
        const int kv0 = (T) * 16;                                                                              \
        f32x4 s = (f32x4){0.f, 0.f, 0.f, 0.f};                                                                 \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) *(uint4*)(kw + i * KROW) = kf[i].u;                      \
        if (!(LAST)) DA_LOAD_K((T) + WAVES);                                                                   \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");                                                 \
        __builtin_amdgcn_wave_barrier();                                                                       \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");                                                 \
        _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) {                                                     \
            Frag ka_, qb_;                                                                                     \
            ka_.u = *(const uint4*)(kr + 64 * ks);                                                             \
            qb_.u = *(const uint4*)(qr + 64 * ks);                                                             \
            s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ka_.v, qb_.v, s, 0, 0, 0);                             \
        }                                                                                                      \
        __builtin_amdgcn_wave_barrier();   /* the next tile's rows are written after these reads */            \
        float sv[4], tmax = -INFINITY;                                                                         \
        _Pragma("unroll") for (int r = 0; r < 4; ++r) {                                                        \
            sv[r] = (!(LAST) || kv0 + 4 * lg + r < len) ? s[r] * sl2 : -INFINITY;                              \
            tmax = fmaxf(tmax, sv[r]);                                                                         \
        }                                                                                                      \
        tmax = rows_max(tmax);                                                                                 \
        const float m_new = fmaxf(m, tmax);                                                                    \
        const float alpha = __builtin_amdgcn_exp2f(m - m_new);                                                 \
        float pr[4], ps = 0.f;                                                                                 \
        _Pragma("unroll") for (int r = 0; r < 4; ++r) {                                                        \
            pr[r] = __builtin_amdgcn_exp2f(sv[r] - m_new);                                                     \
            ps += pr[r];                                                                                       \
        }                                                                                                      \
        ps = rows_sum(ps);                                                                                     \
        l = l * alpha + ps;                                                                                    \
        m = m_new;                                                                                             \
        /* lane (q, lg) holds the probabilities of keys 4 lg .. 4 lg + 3 for query q: the B fragment of P^T */  \
        union { s16x4 v; uint2 u; } pb_;                                                                       \
        pb_.u = make_uint2(pack_bf16x2(pr[0], pr[1]), pack_bf16x2(pr[2], pr[3]));                              \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                        \
            /* rows past the last key are uninitialised memory: a NaN there would survive its zero weight */   \
            const bool keep_ = !(LAST) || kv0 + 4 * lg + i < len;                                              \
            *(uint4*)(vw + i * VROW) = keep_ ? vf[i].u : make_uint4(0u, 0u, 0u, 0u);                           \
        }                                                                                                      \
        if (!(LAST)) DA_LOAD_V((T) + WAVES);                                                                   \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");                                                 \
        __builtin_amdgcn_wave_barrier();                                                                       \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");                                                 \
        if (__builtin_amdgcn_ballot_w64(alpha != 1.0f) != 0) {   /* wave-uniform: the maximum rarely moves */   \
            _Pragma("unroll") for (int dt = 0; dt < D / 16; ++dt) o[dt] *= alpha;                              \
        }                                                                                                      \
        _Pragma("unroll") for (int dt = 0; dt < D / 16; ++dt) {                                                \
            const s16x4 va_ = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(vr + 32 * dt)); \
            o[dt] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(va_, pb_.v, o[dt], 0, 0, 0);                     \
        }                                                                                                      \
        __builtin_amdgcn_wave_barrier();                                                                       \
    }
    const int ntiles = (len + 15) >> 4;
    Frag kf[4], vf[4];
    int t = wave;
    if (t < ntiles) {   // in flight while the queries are prepared
        DA_LOAD_K(t);
        DA_LOAD_V(t);
    }
    if constexpr (FUSED) {
        // One wave per head of this step's projections: GROUP query heads, then the key, then the value.
        const DecodeRopeIn& r = p.rope;
        const int heads = p.kv_heads * GROUP, pos = len;
        const RopeTables tabs{r.cos_tab, r.sin_tab, r.half, r.interleaved, r.max_pos};
        for (int sl = wave; sl < GROUP + 2; sl += WAVES) {
            const int col = sl < GROUP ? (kvh * GROUP + sl) * D : (sl == GROUP ? (heads + kvh) * D : (heads + p.kv_heads + kvh) * D);
            const int64_t src = (int64_t)row * r.ld + col + 2 * lane;
            float x0, x1;
            uint32_t raw;
            if (r.slab) {
                float2 acc = make_float2(0.f, 0.f);
                for (int z = 0; z < r.nslab; ++z) {
                    const float2 tt = *(const float2*)(r.slab + z * r.slab_stride + src);
                    acc.x += tt.x;
                    acc.y += tt.y;
                }
                x0 = acc.x; x1 = acc.y;
                raw = pack_bf16x2(x0, x1);
            } else {
                raw = *(const uint32_t*)(r.qkv + src);
                x0 = __uint_as_float(raw << 16);
                x1 = __uint_as_float(raw & 0xffff0000u);
            }
            if (sl == GROUP + 1) {  // V: cached as is
                *(uint32_t*)(const_cast<bf16_t*>(Vb) + (int64_t)pos * D + 2 * lane) = raw;
                snew[1][2 * lane] = __uint_as_float(raw << 16);
                snew[1][2 * lane + 1] = __uint_as_float(raw & 0xffff0000u);
            } else {
                const float2 y = rope_head(x0, x1, lane, sl < GROUP ? r.q_norm_w : r.k_norm_w, r.eps, tabs, pos);
                const uint32_t packed = pack_bf16x2(y.x, y.y);
                if (sl < GROUP) {
                    *(uint32_t*)(&sq[sl][2 * lane]) = packed;
                } else {
                    *(uint32_t*)(const_cast<bf16_t*>(Kb) + (int64_t)pos * D + 2 * lane) = packed;
                    snew[0][2 * lane] = __uint_as_float(packed << 16);
                    snew[0][2 * lane + 1] = __uint_as_float(packed & 0xffff0000u);
                }
            }
        }
        __syncthreads();
    } else {
        for (int idx = tid; idx < GROUP * D / 8; idx += 64 * WAVES) {
            const int q = idx / (D / 8), ch = idx % (D / 8);
            *(uint4*)(&sq[q][8 * ch]) = *(const uint4*)(p.Q + (int64_t)row * p.q_ld + (kvh * GROUP + q) * D + 8 * ch);
        }
        __syncthreads();
    }
    // Every tile with a successor sits in a block whose loads are unconditional: the compiler counts outstanding loads
    // per path and, where a load is conditional, waits as if it had not been issued, i.e. also for the rows that were
    // just requested.  The last tile of a wave has its own copy of the code.
    if (t < ntiles) {
        for (; t + WAVES < ntiles; t += WAVES) DA_TILE(t, false);   // a tile with a successor is full: no key of it lies past len
        DA_TILE(t, true);
    }
#undef DA_LOAD_K
#undef DA_LOAD_V
#undef DA_TILE
    if constexpr (FUSED) {
        if (wave == 0) {  // the new key and value, from LDS: one more online-softmax step per query head
            const float k0 = snew[0][2 * lane], k1 = snew[0][2 * lane + 1];
#pragma unroll
            for (int q = 0; q < GROUP; ++q) {
                const uint32_t qq = *(const uint32_t*)(&sq[q][2 * lane]);
                const float sc = wave_sum(__uint_as_float(qq << 16) * k0 + __uint_as_float(qq & 0xffff0000u) * k1) * sl2;
                const float mq = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(m), q));
                const float lq = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(l), q));
                const float m_new = fmaxf(mq, sc);
                const float alpha = mq == -INFINITY ? 0.f : exp2f(mq - m_new);
                const float pr = exp2f(sc - m_new);
                if (li == q) {   // the four lanes that hold this query's accumulators
#pragma unroll
                    for (int dt = 0; dt < D / 16; ++dt) {
                        const f32x4 vn = *(const f32x4*)(&snew[1][16 * dt + 4 * lg]);
                        o[dt] = o[dt] * alpha + pr * vn;
                    }
                    m = m_new;
                    l = lq * alpha + pr;
                }
            }
        }
    }
    if (li < GROUP && lg == 0) {
        sm_m[wave][li] = m;
        sm_l[wave][li] = l;
    }
    if (li < GROUP) {
#pragma unroll
        for (int dt = 0; dt < D / 16; ++dt) *(f32x4*)(&so[wave][li][16 * dt + 4 * lg]) = o[dt];
    }
    __syncthreads();
    for (int idx = tid; idx < GROUP * D; idx += 64 * WAVES) {
        const int q = idx / D, d = idx % D;
        float M = -INFINITY;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) M = fmaxf(M, sm_m[w][q]);
        float L = 0.f, O = 0.f;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) {
            const float f = exp2f(sm_m[w][q] - M);
            L += sm_l[w][q] * f;
            O += so[w][q][d] * f;
        }
        p.O[(int64_t)row * p.o_ld + (kvh * GROUP + q) * D + d] = f32_to_bf16(O / L);
    }
    __syncthreads();  // the merge buffers are reused by the next item
    }
}

}  // namespace

void launch_flash_attn(const FlashParams& p, int head_dim, hipStream_t s, double work) {
    if (p.n_tiles <= 0) return;
    Q3_REQUIRE(head_dim == 64 || head_dim == 128, "flash attention: head_dim must be 64 or 128");
    Q3_REQUIRE(p.q_ld % 8 == 0 && p.kv_ld % 8 == 0 && p.kv_hs % 8 == 0 && p.o_ld % 4 == 0, "flash attention: misaligned strides");
    // grouped-query decoder attention: the query heads of a KV head come in pairs, and a workgroup that computes a
    // pair stages every K/V tile once for both (Q3ASR_FLASH_PAIR=0: one head per workgroup)
    const bool pair = head_dim == 128 && p.causal && p.group % 2 == 0 && p.heads % 2 == 0 && q3_knobs().flash_pair != 0;
    dim3 grid(p.n_tiles, pair ? p.heads / 2 : p.heads), block(256);
    ProfScope ps(head_dim == 64 ? (p.causal ? "flash_kernel<64,true>" : "flash_kernel<64,false>")
                                : (p.causal ? "flash_kernel<128,true>" : "flash_kernel<128,false>"), work, s);
    if (head_dim == 64) {
        if (p.causal) hipLaunchKernelGGL((flash_kernel<64, true, 1>), grid, block, 0, s, p);
        else hipLaunchKernelGGL((flash_kernel<64, false, 1>), grid, block, 0, s, p);
    } else {
        if (pair) hipLaunchKernelGGL((flash_kernel<128, true, 2>), grid, block, 0, s, p);
        else if (p.causal) hipLaunchKernelGGL((flash_kernel<128, true, 1>), grid, block, 0, s, p);
        else hipLaunchKernelGGL((flash_kernel<128, false, 1>), grid, block, 0, s, p);
    }
    HIP_CHECK(hipGetLastError());
}

void launch_decode_attn(const DecodeAttnParams& p, int head_dim, hipStream_t s, double work) {
    if (p.B <= 0) return;
    Q3_REQUIRE(head_dim == 128, "decode attention: head_dim must be 128");
    const int items = p.B * p.kv_heads;
    const int pin_waves = q3_knobs().attn_waves;           // 4 or 8: pins the variant (tests, tuning)
    const bool wide = pin_waves ? pin_waves == 8 : items < 256;   // fewer items than CUs: eight waves per item
    dim3 grid(p.max_blocks > 0 ? std::min(items, p.max_blocks) : items), block(wide ? 512 : 256);
    ProfScope ps("decode_attn_kernel", work, s);
    const bool fused = p.rope.qkv || p.rope.slab;
    if (fused) Q3_REQUIRE(p.len_add == 0 && p.rope.ld % 2 == 0 && p.rope.q_norm_w && p.rope.k_norm_w && p.rope.cos_tab && p.rope.sin_tab,
                          "decode attention: incomplete fused q/k/v input");
#define DA_LAUNCH(G, F)                                                                               \
    if (wide) hipLaunchKernelGGL((decode_attn_kernel<G, F, 8>), grid, block, 0, s, p);                \
    else hipLaunchKernelGGL((decode_attn_kernel<G, F, 4>), grid, block, 0, s, p)
    switch (p.group * 2 + (fused ? 1 : 0)) {
        case 2: DA_LAUNCH(1, false); break;
        case 3: DA_LAUNCH(1, true); break;
        case 4: DA_LAUNCH(2, false); break;
        case 5: DA_LAUNCH(2, true); break;
        case 8: DA_LAUNCH(4, false); break;
        case 9: DA_LAUNCH(4, true); break;
        default: throw Q3Error("decode attention: GQA group must be 1, 2 or 4");
    }
#undef DA_LAUNCH
    HIP_CHECK(hipGetLastError());
}
