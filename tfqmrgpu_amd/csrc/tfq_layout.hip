// Block layout conversion (setMatrix / getMatrix) and the shadow vector generator.
//
// Semantics follow the reference (real-space/tfQMRgpu tfqmrgpu.cu:467-603 set_or_getMatrix and
// tfqmrgpu_linalg.hxx:282-380 transpose_blocks_kernel): three element orders inside a block,
// optional transposition, optional conjugation, A stored transposed.  The reference copies the raw
// user array into place and permutes every block in place through shared memory; here the raw
// bytes are staged and one kernel writes the native blocks directly (out of place), which also
// applies the user-order -> column-sorted block permutation of the X-shaped operators.
// Transposed rectangular blocks are indexed as [nC][nR] (the reference mis-indexes them,
// tfqmrgpu_linalg.hxx:322-328; identical for square blocks).
#include "tfq_device.hpp"

namespace tfq {

// offset of part c of element (r, s) of an nR x nC block in the USER's array
__device__ inline uint32_t user_offset(int layout, bool trans, int nR, int nC, int r, int s, int c) {
    int const rows = trans ? nC : nR, cols = trans ? nR : nC;   // shape of the user's array
    int const i = trans ? s : r, j = trans ? r : s;              // position inside it
    (void)rows;
    switch (layout) {
        case TFQMRGPU_LAYOUT_RRRRIIII: return uint32_t(c * nR * nC + i * cols + j);
        case TFQMRGPU_LAYOUT_RRIIRRII: return uint32_t(i * 2 * cols + c * cols + j);
        default:                       return uint32_t((i * cols + j) * 2 + c);  // RIRIRIRI
    }
}

// direction 0: native[n(ub)] := op(user[ub]) ; direction 1: user[ub] := op^-1(native[n(ub)])
// One work group per USER block ub = firstUser + blockIdx.x; `stage` holds the raw user blocks of the
// current batch (block 0 of the stage is user block firstUser); u2n maps user -> native block index
// (nullptr: identity, used for A and B; X-shaped operators pass the column-sorted permutation).
template <typename R, typename U>   // R: library side, U: caller's side
__global__ __launch_bounds__(256) void k_convert(int direction, R* native, U* stage, uint32_t const* u2n,
    uint32_t firstUser, int nR, int nC, int layout, bool trans, bool conj, int ilv)
{
    uint32_t const ub = firstUser + blockIdx.x;
    uint32_t const nb = u2n ? u2n[ub] : ub;
    int const E = 2 * nR * nC;
    U* ublock = stage + size_t(blockIdx.x) * E;
    R* nblock = native + size_t(nb) * E;
    for (int e = threadIdx.x; e < E; e += 256) {
        int const c = e / (nR * nC), r = (e % (nR * nC)) / nC, s = e % nC;
        uint32_t const uo = user_offset(layout, trans, nR, nC, r, s, c);
        bool const neg = (conj && c);
        int const ne = c * nR * nC + plane_offset(ilv, r, s, nC);   // where the library keeps element (c, r, s)
        if (0 == direction) { R const v = R(ublock[uo]); nblock[ne] = neg ? -v : v; }
        else                { U const v = U(nblock[ne]); ublock[uo] = neg ? -v : v; }
    }
}

void launch_convert(int direction, bool userDbl, bool nativeDbl, void* native, void* stage, uint32_t const* u2n,
    uint32_t firstUser, uint32_t nBlocks, int nR, int nC, int layout, bool trans, bool conj, int ilv, hipStream_t s)
{
    if (0 == nBlocks) return;
    dim3 const g(nBlocks), b(256);
    if (nativeDbl) {
        if (userDbl) k_convert<double, double><<<g, b, 0, s>>>(direction, (double*)native, (double*)stage, u2n, firstUser, nR, nC, layout, trans, conj, ilv);
        else         k_convert<double, float ><<<g, b, 0, s>>>(direction, (double*)native, (float*)stage, u2n, firstUser, nR, nC, layout, trans, conj, ilv);
    } else {
        if (userDbl) k_convert<float, double><<<g, b, 0, s>>>(direction, (float*)native, (double*)stage, u2n, firstUser, nR, nC, layout, trans, conj, ilv);
        else         k_convert<float, float ><<<g, b, 0, s>>>(direction, (float*)native, (float*)stage, u2n, firstUser, nR, nC, layout, trans, conj, ilv);
    }
}

// ---- shadow vector: the counter-based hash of tfq_device.hpp written out (k_dot35 and the GLIBC/user modes read v3) ----
__global__ __launch_bounds__(256) void k_shadow_hash(DevPlan d) {
    uint32_t const chunk = blockIdx.x;
    uint32_t const first = d.chunkFirst[chunk], last = d.chunkFirst[chunk + 1];
    uint64_t const col = uint64_t(uint32_t(d.origCol[d.chunkCol[chunk]]));
    int const E = 2 * d.LM * d.LN;
    for (uint32_t b = first; b < last; ++b) {
        uint64_t const key = shadow_key(uint32_t(col), d.rowI[b]);
        float* v = d.v3 + size_t(b) * E;
        int const P = d.LM * d.LN;
        for (int e = threadIdx.x; e < E; e += 256) {   // e = logical element [Re|Im][row][column], stored where the plan's element order puts it
            int const c = e / P, r = (e % P) / d.LN, q = e % d.LN;
            v[c * P + plane_offset(d.ilv, r, q, d.LN)] = shadow_value(key, c, uint32_t(r), uint32_t(q), uint32_t(d.LN));
        }
    }
}

void launch_shadow_hash(DevPlan const& d, hipStream_t s) {
    k_shadow_hash<<<dim3(d.nChunks), dim3(256), 0, s>>>(d);
}

} // namespace tfq
