// march_host.hip -- host logic shared by the march entry points: grid geometry (the block -> work mappings of
// march_common.h evaluated for a problem size), argument checks, and mvp_march_block_map (the mappings exposed to tests).
#include "march_common.h"

namespace mvp {

// ------------------------------------------------------------------------------------------------
// Grid geometry of the march kernels from (N, H, W, K): fills the fields packet_of_block / prim_of_block read.
int setup_block_map(mvp::MarchParams &p) {
    using namespace mvp;
    p.tiles_x = (p.W + kTile - 1) / kTile;
    p.tiles_y = (p.H + kTile - 1) / kTile;
    // packet slots: an image is ceil(tiles_y / kStripRows) strips of kStripRows * tiles_x slots (packet_of_block)
    const long long strips = (p.tiles_y + kStripRows - 1) / kStripRows, S = (long long)kStripRows * p.tiles_x;
    if (strips * S > 0x3fffffffll) return MVP_ERR_UNSUPPORTED;
    p.chunk = (int)((strips * S + 7) / 8);            // 8 * chunk >= the slots of a whole image
    p.images_whole = p.N - p.N % 8;
    const int R = p.N - p.images_whole;
    // XCDs per shared image (measured, forward ms, F = 8 / 4 / 2 or 1): R = 4 (C3) 0.76 / 0.74 / 0.70, (C4) 1.00 / 0.98 /
    // 0.97; R = 2: 0.36 / 0.36 / -; R = 5: 0.61 / 0.66 / 0.80 (F = 1: three XCDs idle)
    p.band_split = R == 4 ? 2 : R == 2 ? 4 : 8;
    p.band_chunk = (int)(((strips + p.band_split - 1) / p.band_split) * S);  // slots of one XCD's share of an image
    const int rounds = (R * p.band_split + 7) / 8;  // groups of 8 / F images
    const long long blocks = 8ll * p.chunk * p.images_whole + 8ll * p.band_chunk * rounds;
    if (blocks > 0x7fffffffll) return MVP_ERR_UNSUPPORTED;
    p.total_packets = (int)blocks;
    return MVP_OK;
}

// blocks of the primitive-centric backward's grid
long long prim_grid_blocks(const mvp::MarchParams &p) {
    return (long long)p.images_whole * p.K +
           8ll * mvp::prim_band_slots(p.K, p.band_split) * (((p.N - p.images_whole) * p.band_split + 7) / 8);
}

extern "C" int mvp_march_block_map(int N, int H, int W, int K, int kind, int first_block, int count, int *out,
                                   int *total_blocks) {
    using namespace mvp;
    if (N < 0 || H < 0 || W < 0 || K < 0 || (kind != 0 && kind != 1) || first_block < 0 || count < 0) return MVP_ERR_BADARG;
    if (count > 0 && !out) return MVP_ERR_BADARG;
    MarchParams p = {};
    p.N = N, p.H = H, p.W = W, p.K = K;
    const int rc = setup_block_map(p);
    if (rc != MVP_OK) return rc;
    const long long total = kind == 0 ? (long long)p.total_packets : prim_grid_blocks(p);
    if (total > 0x7fffffffll) return MVP_ERR_UNSUPPORTED;
    if (total_blocks) *total_blocks = (int)total;
    for (int i = 0; i < count; ++i) {
        const long long b = (long long)first_block + i;
        int n = -1, u = -1;
        const bool ok = b < total && (kind == 0 ? packet_of_block(p, (int)b, n, u) : prim_of_block(p, (int)b, n, u));
        out[2 * i] = ok ? n : -1, out[2 * i + 1] = ok ? u : -1;
    }
    return MVP_OK;
}

int march_common_checks(bool bwd, mvp::MarchParams &p) {
    using namespace mvp;
    if (p.N < 0 || p.H < 0 || p.W < 0 || p.K < 0) return MVP_ERR_BADARG;
    if ((long long)p.N * p.H * p.W == 0) return 1;  // nothing to do
    if (!(p.stepsize > 0.f) || !(p.stepsize < INFINITY) || !(p.fadeexp > 0.f) || !(p.fadescale == p.fadescale))
        return MVP_ERR_BADARG;
    if (p.K > 0 && (p.TD < 2 || p.TH < 2 || p.TW < 2)) return MVP_ERR_UNSUPPORTED;
    if (p.K >= (1 << 24)) return MVP_ERR_UNSUPPORTED;  // list entries pack k into 24 bits
    // warp-field sampler: cell indices are formed in float (tri_zero_pad), exact below 2^24 cells per grid
    if (p.warp && ((long long)p.TD * p.TH * p.TW >= (1ll << 24) || (long long)p.WD * p.WH * p.WW >= (1ll << 24)))
        return MVP_ERR_UNSUPPORTED;
    if (p.campos) {
        if (bwd || p.raypos || p.raydir || p.tminmax) return MVP_ERR_BADARG;
        if (!p.camrot || !p.focal || !p.princpt) return MVP_ERR_BADARG;
        if (!(p.volradius > 0.f) || !(p.volradius < INFINITY)) return MVP_ERR_BADARG;
        if (p.pixelcoords && ((uintptr_t)p.pixelcoords & 7u)) return MVP_ERR_BADARG;
    } else if (!p.raypos || !p.raydir || !p.tminmax) {
        return MVP_ERR_BADARG;
    }
    if (p.K > 0 && (!p.nodeaabb || !p.primpos || !p.primrot || !p.primscale || !p.tplate)) return MVP_ERR_BADARG;
    if (!aligned16(p.tplate) || (p.tminmax && !aligned16(p.tminmax)) || !aligned16(p.nodeaabb)) return MVP_ERR_BADARG;
    if (p.pl_cap < 0 || (p.pl_cap & 3)) return MVP_ERR_BADARG;  // (lists are read four entries = 32 bytes at a time)
    if (p.rayaux && !aligned16(p.rayaux)) return MVP_ERR_BADARG;
    if (p.pl_list && !aligned16(p.pl_list)) return MVP_ERR_BADARG;
    const int rc_grid = setup_block_map(p);
    if (rc_grid != MVP_OK) return rc_grid;
#ifdef MVP_DEBUG_HOOKS
    {
        const char *e = getenv("MVP_DEBUG_FORCE_DFS");
        p.debug_force_dfs = (e && e[0] == '1') ? 1 : 0;
        const char *f = getenv("MVP_DEBUG_SLOT_SWEEP");
        p.debug_slot_sweep = (f && f[0] == '1') ? 1 : 0;
        const char *g = getenv("MVP_DEBUG_STAGE");
        p.debug_stage = g ? atoi(g) : 0;
    }
#endif
    return MVP_OK;
}

}  // namespace mvp
