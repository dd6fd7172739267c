// mlp_host.h -- host-side launch helpers shared by the forward and backward units of the PointNet MLP (pointnet_mlp.hip,
// pointnet_mlp_backward.hip): tile shapes, LDS sizing / occupancy shaping, operand descriptors, the FULL-tile dispatch macro.
#pragma once
#include "mlp_device.h"

namespace sn {

using TileBig = Tile<64, 64, 2, 2>;     // R large: 64 rows x 64 cols per 256-thread workgroup -> >= 2 workgroups per CU
                                        // at B*N = 32768 rows, so one workgroup's load latency hides under another's MFMAs
using TileSmall = Tile<32, 128, 1, 4>;  // R small (FC head at small batch): 32 rows x 128 cols
using TileW = Tile<64, 64, 2, 2>;       // weight gradient: Co x (Ci+1) output tile

template <class T>
static size_t lds_bytes()
{
    return sizeof(float) * std::max(T::LDS_FLOATS, T::WR * 2 * T::BN);
}

}  // namespace sn

using namespace sn;

extern "C" int sn_linear_stats_blocks(int R);

static ActSrc make_act(const float *z, const float *coef, int rows, int ch, int ones_col = -1)
{
    ActSrc a{};
    a.z = z, a.rows = rows, a.ch = ch, a.ones_col = ones_col;
    a.mode = coef ? ACT_BN_RELU : ACT_NONE;
    a.scale = coef, a.shift = coef ? coef + ch : nullptr;
    return a;
}

// Occupancy shaping.  The dispatcher stacks workgroups on a CU up to its resource limit before moving on, so a grid of
// 512 small workgroups can land 4-deep on half of the 256 CUs (measured: SQ_WAIT_INST_ANY 52 % of wave cycles -- four
// waves per SIMD queueing on one matrix pipe) instead of 2-deep on all of them.  Requesting 160 KB / (workgroups per CU
// the grid needs) of LDS makes exactly that many fit, which spreads the grid evenly.
static size_t shaped_lds(size_t needed, dim3 grid)
{
    const size_t nblk = (size_t)grid.x * grid.y * grid.z;
    const size_t per_cu = std::max<size_t>(1, (nblk + 255) / 256);
    const size_t want = std::min<size_t>(64 * 1024, (160 * 1024) / per_cu);
    return std::max(needed, want > 1024 ? want - 1024 : needed);
}

// ---- dispatch helpers: tile x fast-path x operand modes are template parameters (no control flow around loads) ----
#define SN_LAUNCH_T(KERN, T_, FULL_, GRID, ARGS, ...)                                                              \
    do {                                                                                                           \
        const size_t lds_ = shaped_lds(lds_bytes<T_>(), GRID);                                                     \
        if (FULL_)                                                                                                 \
            hipLaunchKernelGGL((KERN<T_, true, __VA_ARGS__>), GRID, dim3(T_::THREADS), lds_, st, ARGS);            \
        else                                                                                                       \
            hipLaunchKernelGGL((KERN<T_, false, __VA_ARGS__>), GRID, dim3(T_::THREADS), lds_, st, ARGS);           \
    } while (0)

static int device_cus()
{
    static int n = 0;
    if (n == 0) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) == hipSuccess &&
            hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0)
            n = v;
        else
            n = 256;  // MI355X
        (void)hipGetLastError();
    }
    return n;
}
