"""GPU parity tests (run with ``pytest -m gpu`` on an MI355X): HIP path (through the C ABI) vs the CPU oracle.

Tolerances are stated in tests/hip_checks.py: bf16 operands / fp32 accumulate / bf16 activations.
"""
import ctypes
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import hip_checks as C  # noqa: E402
from oracle import ucdir_oracle as O  # noqa: E402
from ucdir_amd.spec import UNetConfig  # noqa: E402

SMALL = UNetConfig(inner_channel=64, channel_mults=(1, 2, 4), res_blocks=1, attn_res=(32,), image_size=128)
SID = UNetConfig(inner_channel=64, channel_mults=(1, 2, 4, 8, 8), res_blocks=2, attn_res=(16,), image_size=128)

OP_TOL = 4e-3        # single operator, bf16-representable inputs
FWD_TOL = 1.7e-2     # full forward (61 GroupNorms deep, bf16 operands AND bf16-stored activations).  A torch emulation of the numerics plan
                     # (bf16 rounding exactly where the kernels round) predicts 1.44e-2 for the full SID configuration; measured over the
                     # samples / noise levels the tests use: 1.0-1.3e-2 (small configuration), 1.43-1.51e-2 (full SID: seeds 21, 22, 31, 77
                     # at levels 0.003 ... 0.95).  The spread between samples (+-3 %) is the sampling noise of an rms over a random-weight
                     # network's rounding errors, so the bound sits 12 % above the largest value seen instead of the 2 % of rounds 1-3
                     # (round-3 verdict).  SURVEY.md §8c proposed 1e-2 for bf16 operands with fp32 activations; storing activations in
                     # bf16 is what the extra 0.4e-2 buys back in HBM traffic
CROP_TOL = 2.0e-2    # a 3 x 32 x 32 crop of the same forward against the REFERENCE's own output: 3,072 values instead of
                     # 196,608, the estimate of the same error is noisier


@pytest.fixture(scope="module")
def sid_net():
    return C.build_net(SID)

def _profile_keys(L, fn):
    """Run fn() with the library's per-launch event profiler on and return {key: launches} of the kernels it dispatched."""
    C.ulib.check(L.ucdir_profile_enable(1))
    try:
        r = fn()
    finally:
        C.ulib.check(L.ucdir_profile_enable(0))
    cap = 64
    keys, ln = (ctypes.c_int32 * cap)(), (ctypes.c_int32 * cap)()
    ms, fl, by = (ctypes.c_double * cap)(), (ctypes.c_double * cap)(), (ctypes.c_double * cap)()
    nr = ctypes.c_int32(0)
    C.ulib.check(L.ucdir_profile_read(cap, keys, ln, ms, fl, by, ctypes.byref(nr), C._st()))
    return r, {int(keys[i]): int(ln[i]) for i in range(nr.value)}




@pytest.mark.parametrize("args", [
    (2, 20, 20, 64, 0, 64, 3, 0, False, False, False),     # plain 3x3, TM=64, ragged tile
    (2, 24, 40, 64, 0, 128, 3, 0, True, True, False),      # GroupNorm fold + swish
    (2, 24, 40, 128, 64, 128, 3, 0, True, True, False),    # cat input (ups blocks)
    (1, 33, 17, 128, 64, 64, 3, 0, True, True, False),     # odd sizes
    (2, 32, 32, 128, 0, 128, 3, 1, False, False, False),   # Downsample
    (2, 16, 16, 128, 0, 128, 3, 2, False, False, False),   # Upsample
    (1, 16, 24, 64, 0, 64, 3, 2, False, False, False),
    (2, 24, 40, 128, 64, 64, 1, 0, False, False, True),    # res_conv 1x1 + residual
    (1, 12, 12, 512, 0, 512, 1, 0, True, False, False),    # qkv-like 1x1 with GN fold
    (1, 18, 18, 512, 512, 512, 3, 0, True, True, False),   # split-K: 16 workgroups x 32 chunks (the B = 1 latency path)
    (1, 36, 36, 256, 0, 512, 3, 0, True, True, True),      # split-K with a residual
    (1, 18, 18, 512, 0, 512, 3, 2, False, False, False),   # split-K Upsample parity launches
    (3, 18, 18, 512, 256, 512, 3, 0, True, True, False),   # split-K, ragged chunk ranges (24 chunks over 5 splits)
])
def test_conv_gemm(args):
    m = C.conv_case(*args)
    assert not m["nan"]
    assert m["rel_rms"] < OP_TOL, m
    assert m["max_abs_border"] < 0.05 * max(m["ref_rms"], 1.0), m     # border classes of the GN fold
    assert m["stats_rel"] < 1e-3, m                                    # GroupNorm partial sums


@pytest.mark.parametrize("args", [
    (3, 32, 48, True, True, 4096),    # 16 x 16 tiles, all border tiles, one tile per workgroup; GroupNorm fold + swish
    (3, 64, 80, True, True, 7),       # persistent ranges of 8-9 tiles crossing sample boundaries (grid forced to 7 workgroups)
    (2, 48, 48, False, False, 5),     # no fold, no activation (bias only)
    (4, 288, 288, True, True, 0),     # the network's level-0 size on one workgroup per CU (1296 tiles: above the engine's 4-tiles-per-CU threshold)
], ids=["even_small", "ranges_cross_samples", "plain", "level0"])
def test_conv_persistent(args):
    """conv_ws_kernel (3x3, 64 -> 64, persistent, weights in registers): tile ranges, sample crossings, border classes, stats."""
    B, H, W, gn, silu, grid = args
    L = C.ulib.load()
    C.ulib.check(L.ucdir_debug_flag(b"persist_grid", grid))
    try:
        m = C.conv_case(B, H, W, 64, 0, 64, 3, 0, gn, silu, False, seed=3)
    finally:
        C.ulib.check(L.ucdir_debug_flag(b"persist_grid", 0))
    assert not m["nan"] and m["rel_rms"] < OP_TOL, m
    assert m["max_abs_border"] < 0.05 * max(m["ref_rms"], 1.0), m
    assert m["stats_rel"] < 1e-3, m


@pytest.mark.parametrize("args", [
    # B, H, W, c0, c1, cout, mode, gn, silu, residual, persist_grid
    (2, 24, 40, 64, 0, 256, 0, True, True, False, 0),       # two row-of-256 units per tile, ragged last tile, GroupNorm fold (all nine border classes)
    (3, 18, 18, 128, 64, 512, 0, True, True, False, 0),     # cat input, two row tiles, tiles spanning samples (400 positions per sample)
    (5, 10, 12, 64, 0, 256, 0, True, False, False, 3),      # 3 workgroups: whole units + a stream-K remainder cut across workgroups, several samples per tile
    (2, 36, 36, 256, 0, 512, 0, True, True, True, 7),       # residual; 7 workgroups: units cut into three parts (finish kernel sums in workgroup order)
    (2, 16, 24, 128, 0, 256, 2, False, False, False, 0),    # Upsample: four parity classes of 2 x 2 taps
    (3, 9, 9, 256, 0, 512, 2, False, False, False, 5),      # Upsample, stream-K across parity classes and row tiles
    (1, 72, 72, 192, 64, 256, 0, True, True, False, 0),     # the concatenation boundary inside the chunk sequence (c0 = 192 = 6 chunks of 8)
    (16, 18, 18, 512, 0, 512, 0, True, True, False, 0),     # the 18^2 level of the bench configuration
    (2, 24, 40, 64, 64, 128, 0, True, True, False, 0),      # C_out = 128: 128-row x 512-position units (eight waves along the positions)
    (3, 20, 28, 128, 0, 128, 0, True, True, True, 3),       # the same with a residual, stream-K over 3 workgroups
    (2, 16, 24, 128, 0, 128, 2, False, False, False, 0),    # Upsample at C_out = 128 (all halo pieces of a chunk at its first sub-step)
    (1, 144, 144, 64, 0, 128, 0, True, True, False, 0),     # the widest level this kernel takes in the network (halo of 806 positions; three strips of 48 for the 4-wave kind)
    (2, 50, 70, 64, 0, 128, 0, True, True, False, 0),       # two strips of 35 columns (4-wave kind), ragged in every direction
    (1, 40, 100, 64, 0, 256, 2, False, False, False, 0),    # Upsample on strips
    (16, 36, 36, 512, 0, 512, 0, True, True, False, 0),     # the 36^2 level of the bench configuration
    (4, 20, 24, 128, 0, 256, 0, True, True, False, 8),      # kind 1: 9 units of 4 chunks on 8 workgroups: a remainder of ONE unit has fewer chunks than workgroups
                                                            # (round-4 advice: workgroups without a range must not contribute stale partial slots)
], ids=["fold_ragged", "cat_samples", "streamk_3wg", "res_streamk_7wg", "up_256", "up_streamk", "cat_chunks", "level4_b16",
        "rows128", "rows128_res_streamk", "rows128_up", "rows128_wide", "strips_ragged", "strips_up", "level3_b16", "short_remainder"])
@pytest.mark.parametrize("kind", [1, 2], ids=["persistent8", "oneshot4"])
def test_conv_stream_k(args, kind):
    """conv_sk_kernel (persistent stream-K 3x3 conv / Upsample parity classes on 256-row x 256-position linear tiles) + its finish
    kernel against torch through the C ABI: batch-flattened tiles (borders computed and dropped, tiles crossing samples), per-sample
    GroupNorm fold in the epilogue, fixed-order partial sums, statistics.  kind 1: one persistent 8-wave workgroup per CU (256 x 256 or
    128 x 512 units, stream-K remainder); kind 2: 4-wave workgroups of 128 x 256, two per CU, one unit each (with a forced grid:
    ranges of units and a stream-K remainder too); both on vertical strips where the image is too wide for the halo buffers."""
    B, H, W, c0, c1, cout, mode, gn, silu, residual, grid = args
    L = C.ulib.load()
    C.ulib.check(L.ucdir_debug_flag(b"convsk", kind))
    C.ulib.check(L.ucdir_debug_flag(b"persist_grid", grid))
    C.ulib.check(L.ucdir_debug_flag(b"skmix", 0))            # the plain kernel (level3_b16 would take the mixed one: its own test below)
    try:
        (m, keys) = _profile_keys(L, lambda: C.conv_case(B, H, W, c0, c1, cout, 3, mode, gn, silu, residual, seed=5))
        m2 = C.conv_case(B, H, W, c0, c1, cout, 3, mode, gn, silu, residual, seed=5)
    finally:
        C.ulib.check(L.ucdir_debug_flag(b"skmix", -1))
        C.ulib.check(L.ucdir_debug_flag(b"persist_grid", 0))
        C.ulib.check(L.ucdir_debug_flag(b"convsk", -1))
    assert (125 if kind == 1 else 127) + (1 if mode == 2 else 0) in keys, keys      # the new kernel ran, not a fallback
    assert not m["nan"] and m["rel_rms"] < OP_TOL, m
    assert m["max_abs_border"] < 0.05 * max(m["ref_rms"], 1.0), m
    assert m["stats_rel"] < 1e-3, m
    assert m2["rel_rms"] == m["rel_rms"] and m2["max_abs"] == m["max_abs"] and m2["stats_rel"] == m["stats_rel"], (m, m2)   # run to run


@pytest.mark.parametrize("args", [
    (2, 24, 40, 128, 64, 128),       # 192 -> 128 with its res_conv: one row tile
    (3, 18, 18, 512, 256, 512),      # four row tiles, tiles spanning samples
    (1, 72, 72, 256, 128, 256),      # two strips of 36 columns
    (2, 30, 44, 64, 0, 128),         # no concatenation, ragged
    (16, 18, 18, 512, 512, 512),     # the bench configuration's 18^2 block: 100 units for 256 CUs -> every unit's K cut in two (finish kernel), res_conv still in the launch
], ids=["rows128", "rows512_samples", "strips", "plain_ragged", "level4_ksplit"])
def test_conv_stream_k_with_res_conv(args):
    """conv1 (GroupNorm fold + swish) with the block's 1x1 res_conv as the LAST workgroups of the same conv_sk launch (4-wave kind):
    both outputs against torch, statistics of conv1's output, run to run bit-identical."""
    B, H, W, c0, c1, cout = args
    L = C.ulib.load()
    C.ulib.check(L.ucdir_debug_flag(b"convsk", 2))
    try:
        (m, keys) = _profile_keys(L, lambda: C.conv_res_case(B, H, W, c0, c1, cout, seed=7))
        m2 = C.conv_res_case(B, H, W, c0, c1, cout, seed=7)
    finally:
        C.ulib.check(L.ucdir_debug_flag(b"convsk", -1))
    assert 127 in keys and 100 not in keys and 0 not in keys, keys     # one launch: no separate 1x1 GEMM
    assert not m["nan"] and m["rel_rms"] < OP_TOL and not m["res_nan"] and m["res_rel_rms"] < OP_TOL, m
    assert m["max_abs_border"] < 0.05 * max(m["ref_rms"], 1.0) and m["stats_rel"] < 1e-3, m
    assert m2 == m, (m, m2)


@pytest.mark.parametrize("args", [
    # B, H, W, c0, c1, cout, residual, with res_conv, forced
    (2, 24, 40, 64, 0, 256, False, False, 1),        # two row tiles: 4 wide tiles + narrow tiles behind them, ragged end, all nine border classes
    (3, 18, 18, 128, 64, 512, False, True, 1),       # cat input, four row tiles, tiles spanning samples, the block's res_conv units behind the short ones
    (2, 36, 36, 256, 0, 512, True, False, 1),        # residual
    (2, 50, 70, 64, 0, 128, False, False, 1),        # one row tile (8 wide tiles, short tiles in multiples of 4), two strips, tiles past the end of the space
    (16, 36, 36, 512, 0, 512, False, False, -1),     # the bench configuration's 36^2 level: engages by its own occupancy rule (344 units on 512 slots)
    (16, 36, 36, 512, 256, 512, False, True, -1),    # ... with the res_conv tail
], ids=["fold_ragged", "cat_res_conv", "residual", "strips_rows128", "level3_b16", "level3_b16_res_conv"])
def test_conv_sk_mixed_wide_and_short_units(args):
    """conv_sk_kernel<1, 4, 9> with a mixed schedule (round 6): wide units (128 rows x 256 positions, wave tile 128 x 64) over the first
    pixel tiles and SHORT units (64 rows x 256 positions, wave tile 64 x 64, two per 128-row tile) over the rest of ONE launch against
    torch - halo positions behind the end of the space wrap to sample 0's border - incl. the res_conv tail; profiler key 129 = the
    mixed schedule ran; run to run bit-identical."""
    B, H, W, c0, c1, cout, residual, with_res, force = args
    L = C.ulib.load()
    C.ulib.check(L.ucdir_debug_flag(b"convsk", 2))
    C.ulib.check(L.ucdir_debug_flag(b"skmix", force))
    try:
        if with_res:
            (m, keys) = _profile_keys(L, lambda: C.conv_res_case(B, H, W, c0, c1, cout, seed=7))
            m2 = C.conv_res_case(B, H, W, c0, c1, cout, seed=7)
        else:
            (m, keys) = _profile_keys(L, lambda: C.conv_case(B, H, W, c0, c1, cout, 3, 0, True, True, residual, seed=5))
            m2 = C.conv_case(B, H, W, c0, c1, cout, 3, 0, True, True, residual, seed=5)
    finally:
        C.ulib.check(L.ucdir_debug_flag(b"skmix", -1))
        C.ulib.check(L.ucdir_debug_flag(b"convsk", -1))
    assert 129 in keys and 127 not in keys and 100 not in keys, keys
    assert not m["nan"] and m["rel_rms"] < OP_TOL, m
    if with_res:
        assert not m["res_nan"] and m["res_rel_rms"] < OP_TOL, m
    assert m["max_abs_border"] < 0.05 * max(m["ref_rms"], 1.0) and m["stats_rel"] < 1e-3, m
    assert m2 == m, (m, m2)


@pytest.mark.parametrize("args", [
    (3, 32, 48, 64, 64, 64, 4096),     # conv_ws128: 8 x 16 tiles, one tile per workgroup, all border tiles
    (3, 64, 80, 64, 64, 64, 7),        # persistent ranges crossing sample boundaries (grid forced to 7 workgroups)
    (2, 288, 288, 64, 64, 64, 0),      # the network's level-0 size (ups.17 / ups.18)
    (2, 40, 56, 128, 64, 64, 0),       # 192 -> 64: conv3x3_halo<64, true> (10th tap)
    (2, 24, 40, 128, 64, 128, 0),      # 192 -> 128: res_conv as tail workgroups of conv3x3_halo<128>
], ids=["ws128_small", "ws128_ranges", "ws128_level0", "tap10_192", "tail_128"])
def test_conv_with_fused_res_conv(args):
    """conv1 + the block's res_conv in one launch (the UNet's ups blocks), every kernel that implements it."""
    B, H, W, c0, c1, cout, grid = args
    L = C.ulib.load()
    C.ulib.check(L.ucdir_debug_flag(b"persist_grid", grid))
    try:
        m = C.conv_res_case(B, H, W, c0, c1, cout, seed=4)
    finally:
        C.ulib.check(L.ucdir_debug_flag(b"persist_grid", 0))
    assert not m["nan"] and not m["res_nan"] and m["rel_rms"] < OP_TOL and m["res_rel_rms"] < OP_TOL, m
    assert m["max_abs_border"] < 0.05 * max(m["ref_rms"], 1.0), m
    assert m["stats_rel"] < 1e-3, m


@pytest.mark.parametrize("Cc", [64, 128, 256, 512])
def test_akgm(Cc):
    m = C.akgm_case(2, Cc, 20, 24)
    assert not m["nan"] and m["rel_rms"] < OP_TOL, m
    assert m["max_abs_border"] < 0.06, m


@pytest.mark.parametrize("args", [
    (3, 64, 32, 48, 4096),   # 16 x 16 tiles, every tile a border tile, one tile per workgroup (grid >= tiles)
    (3, 64, 64, 80, 7),      # persistent ranges of 8-9 tiles that cross sample boundaries (grid forced to 7 workgroups)
    (2, 64, 40, 56, 3),      # ragged tiles (clamped halo, masked stores) inside multi-tile ranges
    (4, 64, 288, 288, 0),    # the network's level-0 size: 1296 tiles on one workgroup per CU (above the 4-tiles-per-CU threshold)
    (3, 128, 32, 48, 4096),  # 16 channels per group: one 64-channel plane per workgroup, one tile per workgroup pair
    (3, 128, 64, 80, 14),    # ... ranges of 8-9 tiles that cross sample boundaries (7 workgroup pairs)
    (8, 128, 144, 144, 0),   # ... the network's 144^2 level on one workgroup per CU
    (3, 256, 32, 48, 4096),  # 32 channels per group (akgm_ws32_kernel): one group per workgroup, 32 x 8 tiles, one tile per workgroup
    (2, 256, 48, 40, 16),    # ... 24 x 8 tiles, ranges of 10 tiles that cross the sample boundary (two workgroups per group)
    (5, 256, 72, 72, 0),     # ... the network's 72^2 level on one workgroup per CU
    (2, 512, 20, 24, 16),    # 64 channels per group (akgm_ws64_kernel): one half group per workgroup, linear tiles of 128 positions, one range per role that crosses the sample
    (3, 512, 18, 18, 32),    # ... two ranges of 4-5 tiles per role that cross sample boundaries
    (2, 512, 12, 60, 48),    # ... a wide plane: the halo of a 128-position tile at its LDS limit (254 of 272 positions), three ranges per role
    (6, 512, 36, 36, 0),     # ... the network's 36^2 level on one workgroup per CU (128-position tiles)
    (8, 512, 18, 18, 0),     # ... the network's 18^2 level (64-position tiles: fewer than four 128-position tiles per range)
], ids=["even_small", "ranges_cross_samples", "ragged_ranges", "level0", "cg16_small", "cg16_ranges", "cg16_level1",
        "cg32_small", "cg32_ranges", "cg32_level2", "cg64_small", "cg64_ranges", "cg64_wide", "cg64_level3", "cg64_level4"])
def test_akgm_persistent(args):
    """akgm_ws_kernel / akgm_ws32_kernel / akgm_ws64_kernel (8 | 16 | 32 | 64 channels per group, persistent, weight-stationary):
    tile ranges, sample crossings, border classes; the profiler's key proves which kernel ran."""
    B, Cc, H, W, grid = args
    L = C.ulib.load()
    C.ulib.check(L.ucdir_debug_flag(b"persist_grid", grid))
    try:
        m, keys = _profile_keys(L, lambda: C.akgm_case(B, Cc, H, W, seed=5))
        m2 = C.akgm_case(B, Cc, H, W, seed=5)
    finally:
        C.ulib.check(L.ucdir_debug_flag(b"persist_grid", 0))
    if Cc == 512:
        assert 116 in keys, keys
    assert not m["nan"] and m["rel_rms"] < OP_TOL, m
    assert m["max_abs_border"] < 0.06, m
    assert m["stats_rel"] < 1e-3, m                                      # GroupNorm partial sums of the output
    assert m["max_abs"] == m2["max_abs"] and m["rel_rms"] == m2["rel_rms"] and m["stats"] == m2["stats"]     # reproducible


@pytest.mark.parametrize("args", [(3, 64, 64, 80, 6), (3, 128, 64, 80, 12), (2, 64, 40, 56, 2)], ids=["cg8", "cg16", "cg8_th8"])
def test_akgm_block_kernel_at_narrow_groups(args):
    """akgm_ws32_kernel<8 | 16> (the 32-feature-block kernel templated on the group width; not the default at these widths)."""
    B, Cc, H, W, grid = args
    L = C.ulib.load()
    C.ulib.check(L.ucdir_debug_flag(b"wsb", 1))
    C.ulib.check(L.ucdir_debug_flag(b"persist_grid", grid))
    try:
        m = C.akgm_case(B, Cc, H, W, seed=9)
        m2 = C.akgm_case(B, Cc, H, W, seed=9)
    finally:
        C.ulib.check(L.ucdir_debug_flag(b"persist_grid", 0))
        C.ulib.check(L.ucdir_debug_flag(b"wsb", -1))
    assert not m["nan"] and m["rel_rms"] < OP_TOL, m
    assert m["stats_rel"] < 1e-3, m
    assert m["max_abs"] == m2["max_abs"] and m["rel_rms"] == m2["rel_rms"] and m["stats"] == m2["stats"]


@pytest.mark.parametrize("shape", [(2, 128, 12, 10), (1, 512, 36, 36), (1, 512, 18, 18), (3, 256, 20, 24), (5, 512, 18, 18)])
@pytest.mark.parametrize("flash", [1, -1], ids=["flash", "engine_choice"])
def test_attention(shape, flash):
    m = C.attention_case(*shape, flash=flash)
    assert not m["nan"] and m["rel_rms_branch"] < 1.2e-2, m


@pytest.mark.parametrize("shape", [(2, 64, 96), (1, 256, 256), (1, 40, 72)])
def test_predictor(shape):
    m = C.predictor_case(*shape)
    assert not m["nan"] and m["rel_rms"] < 1.5e-2, m


def test_sampler_step_exact():
    for k, m in C.sampler_step_case().items():
        assert m["max_abs"] < 2e-6, (k, m)       # fp32 point-wise; differences are FMA contraction only


def test_in_kernel_noise_is_standard_normal_and_counter_based():
    """sampler_step_rng / fill_normal (Philox4x32-10 + Box-Muller in the update kernel): N(0, 1) moments over 4 M samples,
    no correlation between neighbours / steps / seeds, a pure function of (seed, step, element) - independent of how the tensor
    is split - and the update equals the plain kernel fed with the same noise."""
    from ucdir_amd.ucdir import fill_normal_, sampler_step_, sampler_step_rng_
    n = 1 << 22
    a = fill_normal_(torch.empty(n, device="cuda"), 1234, 3)
    b = fill_normal_(torch.empty(n, device="cuda"), 1234, 3)
    assert torch.equal(a, b)
    x = a.double()
    m, v = float(x.mean()), float(x.var())
    sk, ku = float((x ** 3).mean()), float((x ** 4).mean())
    assert abs(m) < 2.5e-3 and abs(v - 1) < 4e-3 and abs(sk) < 1e-2 and abs(ku - 3) < 3e-2, (m, v, sk, ku)
    assert float(x.abs().max()) < 6.5                                       # 24-bit uniforms: |z| <= sqrt(2 ln 2^25) = 5.9
    other_step = fill_normal_(torch.empty(n, device="cuda"), 1234, 4).double()
    other_seed = fill_normal_(torch.empty(n, device="cuda"), 1235, 3).double()
    for y in (other_step, other_seed, x.roll(1), x.roll(2), x.roll(4)):
        assert abs(float((x * y).mean())) < 2.5e-3
    # counter-based: the second half of a tensor is the second half of the stream, whatever the launch
    part = fill_normal_(torch.empty(n // 2, device="cuda"), 1234, 3)
    assert torch.equal(part, a[:n // 2])
    # tail elements (n not a multiple of 4)
    t = fill_normal_(torch.empty(1027, device="cuda"), 9, 1)
    assert torch.equal(t[:1024], fill_normal_(torch.empty(1024, device="cuda"), 9, 1)) and bool(torch.isfinite(t).all())
    # the fused update == the plain update with the same noise
    g = torch.Generator().manual_seed(0)
    xt = torch.randn(2, 3, 64, 64, generator=g).cuda(); eps = torch.randn(2, 3, 64, 64, generator=g).cuda()
    z = fill_normal_(torch.empty_like(xt), 77, 5)
    r1 = sampler_step_(xt.clone(), eps, z, 1.7, 1.3, 0.4, 0.6, 0.25)
    r2 = sampler_step_rng_(xt.clone(), eps, 77, 5, 1.7, 1.3, 0.4, 0.6, 0.25)
    assert float((r1 - r2).abs().max()) < 2e-6
    r3 = sampler_step_rng_(xt.clone(), eps, 77, 5, 1.7, 1.3, 0.4, 0.6, 0.0)       # last step: sigma = 0, no noise
    r4 = sampler_step_(xt.clone(), eps, None, 1.7, 1.3, 0.4, 0.6, 0.0)
    assert float((r3 - r4).abs().max()) < 2e-6


def test_forward_small_vs_oracle_and_golden(golden_dir):
    out, eps, ref = C.forward_case(SMALL, 2, 64, 48, [0.0029, 0.6], seed=11, taps=True)
    assert out["eps"]["rel_rms"] < FWD_TOL, out["eps"]
    for k, m in out.items():
        assert not m["nan"] and m["rel_rms"] < FWD_TOL, (k, m)
    g = np.load(os.path.join(golden_dir, "small_forward.npz"))       # output of the real reference
    gm = C.metrics(eps, torch.from_numpy(g["eps"].astype(np.float32)))
    print("SMALL forward vs oracle:", out["eps"], " vs reference golden:", gm, " worst layer:",
          max((m["rel_rms"], k) for k, m in out.items()))
    assert gm["rel_rms"] < FWD_TOL, gm


@pytest.mark.parametrize("i", [0, 1, 2], ids=["t49", "t25", "t0"])
def test_forward_sid_full_config(golden_dir, sid_net, i):
    """All three stored levels (t = 49: level 0.0029, where the sampler amplifies eps errors 349x; t = 25; t = 0) against the
    oracle AND the crop / 8x-subsampled grid / statistics of the real reference's output."""
    g = np.load(os.path.join(golden_dir, "sid_forward.npz"))
    out, eps, ref = C.forward_case(SID, 1, 256, 256, [float(g["levels"][i])], seed=21, taps=False, net_sd=sid_net)
    assert out["eps"]["rel_rms"] < FWD_TOL, out["eps"]
    gm = C.metrics(eps[0, :, 100:132, 60:92], torch.from_numpy(g[f"eps{i}_crop"]))
    gd = C.metrics(eps[0, :, ::8, ::8], torch.from_numpy(g[f"eps{i}_ds"]))
    print("full SID forward level", i, "vs oracle:", out["eps"], " crop vs reference golden:", gm, " subsampled:", gd)
    assert gm["rel_rms"] < CROP_TOL and gd["rel_rms"] < CROP_TOL, (gm, gd)
    st = np.array([eps.mean(), eps.std(), eps.min(), eps.max()], dtype=np.float64)
    assert abs(st[0] - g[f"eps{i}_stats"][0]) < 0.02 * g[f"eps{i}_stats"][1] and abs(st[1] / g[f"eps{i}_stats"][1] - 1) < 0.01, (st, g[f"eps{i}_stats"])


def test_forward_sid_full_config_b2(golden_dir, sid_net):
    """SURVEY 8c fixture (ii), B = 2: both samples of one launch against the reference's own output."""
    g = np.load(os.path.join(golden_dir, "sid_forward_b2.npz"))
    out, eps, ref = C.forward_case(SID, 2, 256, 256, [float(v) for v in g["levels"].reshape(-1)], seed=22, taps=False, net_sd=sid_net)
    assert out["eps"]["rel_rms"] < FWD_TOL, out["eps"]
    for b in range(2):
        gm = C.metrics(eps[b, :, 100:132, 60:92], torch.from_numpy(g[f"b{b}_crop"]))
        gd = C.metrics(eps[b, :, ::8, ::8], torch.from_numpy(g[f"b{b}_ds"]))
        assert gm["rel_rms"] < CROP_TOL and gd["rel_rms"] < CROP_TOL, (b, gm, gd)


def test_real_image_through_predictor_and_denoiser(golden_dir, sid_net):
    """SURVEY 8c fixture (v): a 256^2 crop of the reference's own sample image as the condition; the HIP predictor's output
    is the guide of the HIP denoiser (ResiGaussianGuideDY.super_resolution's wiring); both against the reference's outputs."""
    g = np.load(os.path.join(golden_dir, "sid_real_image.npz"))
    net, sd = sid_net
    from ucdir_amd.weights import synth_inputs
    dev = torch.device("cuda")
    c = (torch.from_numpy(g["cond_u8"]).permute(2, 0, 1)[None].float() / 255.0 * 2.0 - 1.0).to(dev)
    xt = torch.from_numpy(synth_inputs(1, 256, 256, seed=23)[2]).to(dev)
    with torch.no_grad():
        pred = net.predictor(c)
        eps = net.denoise_fn(torch.cat([c, xt], 1), torch.from_numpy(g["level"]).to(dev), pred).cpu()
    pm = C.metrics(pred.cpu()[0, :, 100:132, 60:92], torch.from_numpy(g["pred_crop"]))
    gm = C.metrics(eps[0, :, 100:132, 60:92], torch.from_numpy(g["eps_crop"]))
    gd = C.metrics(eps[0, :, ::8, ::8], torch.from_numpy(g["eps_ds"]))
    print("real image: predictor", pm, " eps crop", gm, " eps subsampled", gd)
    assert pm["rel_rms"] < 1.5e-2 and gm["rel_rms"] < CROP_TOL + 5e-3 and gd["rel_rms"] < CROP_TOL + 5e-3, (pm, gm, gd)   # the guide itself carries the predictor's bf16 error


def test_time_embedding_direct(sid_net):
    """PositionalEncoding + noise_level_mlp + every block's noise_func (model/ucdir.py:24-29, 106, 125, 212-214) as computed by
    time_mlp_kernel (fp32), read back per block, against the oracle - over the whole range of noise levels the schedule produces."""
    net, sd = sid_net
    from ucdir_amd.spec import unet_layers
    from ucdir_amd.weights import synth_inputs
    B = 4
    cond, guide, x_t = map(torch.from_numpy, synth_inputs(B, 64, 64, seed=2))
    lvl = torch.tensor([[1e-4], [0.03], [0.5], [0.9999]])
    with torch.no_grad():
        net.denoise_fn(torch.cat([cond, x_t], 1).cuda(), lvl.cuda(), guide.cuda())
    temb = O.noise_embedding(sd, lvl, "denoise_fn.")
    worst, n = 0.0, 0
    for Ld in unet_layers(net.denoise_fn.cfg):
        if Ld.kind != "block":
            continue
        got = net.denoise_fn.debug_read(Ld.name, "attw").cpu()
        ref = O.time_weights(sd, "denoise_fn." + Ld.name + ".res_block.", temb)
        assert got.shape == ref.shape == (B, 8)
        worst = max(worst, float((got - ref).abs().max() / ref.abs().max()))
        n += 1
    assert n == 27 and worst < 2e-5, (n, worst)      # fp32 both sides: __sinf / __expf vs libm


def test_forward_batch_is_independent(sid_net):
    """B samples in one launch == B single launches (per-sample GroupNorm statistics)."""
    net, sd = sid_net
    from ucdir_amd.weights import synth_inputs
    cond, guide, x_t = map(torch.from_numpy, synth_inputs(3, 96, 64, seed=3))
    lvl = torch.tensor([[0.1], [0.5], [0.9]])
    L = C.ulib.load()
    C.ulib.check(L.ucdir_debug_flag(b"splitk", 0))       # split-K depends on the grid size, i.e. on B: another summation order
    try:
        with torch.no_grad():
            full = net.denoise_fn(torch.cat([cond, x_t], 1).cuda(), lvl.cuda(), guide.cuda()).cpu()
            one = net.denoise_fn(torch.cat([cond[1:2], x_t[1:2]], 1).cuda(), lvl[1:2].cuda(), guide[1:2].cuda()).cpu()
    finally:
        C.ulib.check(L.ucdir_debug_flag(b"splitk", -1))
    assert torch.equal(full[1:2], one)          # bit exact: same tiles, same reduction order
    with torch.no_grad():                       # default launch configuration (split-K for the single sample's small grids)
        one_s = net.denoise_fn(torch.cat([cond[1:2], x_t[1:2]], 1).cuda(), lvl[1:2].cuda(), guide[1:2].cuda()).cpu()
        one_s2 = net.denoise_fn(torch.cat([cond[1:2], x_t[1:2]], 1).cuda(), lvl[1:2].cuda(), guide[1:2].cuda()).cpu()
    assert torch.equal(one_s, one_s2)           # partial tiles are summed in a fixed order: reproducible
    m = C.metrics(one_s, one)
    assert m["rel_rms"] < FWD_TOL, m


def test_forward_bit_reproducible_at_bench_size(sid_net):
    """BASELINE configs[1] size (B = 16, 256^2 -> 288^2 compute, ~10^4 workgroups per launch): two forwards of the same
    inputs are bit-identical although every GroupNorm statistic is accumulated with atomics in arrival order — the
    accumulators are fixed-point integers (csrc/common.h stat_add) — and the result is finite and sample-dependent."""
    net, sd = sid_net
    from ucdir_amd.weights import synth_inputs
    cond, guide, x_t = map(torch.from_numpy, synth_inputs(16, 256, 256, seed=4))
    lvl = torch.linspace(0.01, 0.99, 16).reshape(16, 1)
    x6 = torch.cat([cond, x_t], 1).cuda()
    with torch.no_grad():
        a = net.denoise_fn(x6, lvl.cuda(), guide.cuda()).clone()
        b = net.denoise_fn(x6, lvl.cuda(), guide.cuda())
    assert torch.equal(a, b)
    assert bool(torch.isfinite(a).all()) and a.shape == (16, 3, 256, 256)
    assert float((a[0] - a[1]).abs().max()) > 1e-3
    # sample 5 launched alone: small grids switch some layers to 64-row tiles (engine.hip run_conv), i.e. another fp32
    # summation order -> another realisation of the bf16 rounding noise, which this random-weight network amplifies
    # to the size of the build-vs-oracle error itself (measured 1.1e-2); bit-exactness holds when the tilings agree
    # (test_forward_batch_is_independent)
    with torch.no_grad():
        one = net.denoise_fn(x6[5:6].contiguous(), lvl[5:6].cuda(), guide[5:6].cuda())
    m = C.metrics(a[5:6], one.cpu())
    assert m["rel_rms"] < FWD_TOL, m



EMU_LAYER_TOL = 2e-3   # one layer of the HIP path against the oracle's bf16-emulation mode ON THE SAME INPUTS (teacher forcing): what is left is
                       # fp32 summation order and single bf16 rounding flips.  Measured (tools/_emu_probe.py): worst layer 6.7e-4 (full SID, B = 1),
                       # 4.4e-4 (B = 4), 8.9e-4 (small configuration) - the attention blocks of the 18^2 / 36^2 levels; everything else <= 4e-4


@pytest.mark.parametrize("B", [1, 4, 16])
def test_full_sid_forward_layer_by_layer_vs_bf16_emulation(sid_net, B):
    """The second oracle mode (round-4 verdict).  End to end the emulation is one more realisation of the rounding noise: it sits
    1.50e-2 from the fp32 oracle - exactly where the HIP path sits (1.47 - 1.50e-2), which shows that the whole build-vs-oracle error IS
    the numerics plan - and 1.2e-2 from the HIP path, so end to end it is no tighter a net than FWD_TOL.  Layer by layer on identical
    inputs it is: every stored activation (36 layer outputs + 27 h1 tensors) of a full SID forward against the emulated layer fed with
    the HIP path's own input activations must agree to EMU_LAYER_TOL, an order of magnitude below FWD_TOL - a systematic error of a
    few 1e-3 in any single kernel shows.  B = 1: the one-shot / split-K kernels; B = 4, 16: the persistent kernels and conv_sk
    (B = 16: the dispatch bench.py times; samples 0 and 15 are checked)."""
    if B == 16:
        # the CPU emulation of 16 samples would take minutes: check the batch's first and last sample through B = 16 HIP forwards
        net, sd = sid_net
        from ucdir_amd.weights import synth_inputs
        import torch.nn.functional as F
        from ucdir_amd.spec import unet_layers
        cond, guide, x_t = map(torch.from_numpy, synth_inputs(16, 256, 256, seed=41))
        lvl = torch.linspace(0.02, 0.97, 16).reshape(16, 1)
        x6 = torch.cat([cond, x_t], 1)
        with torch.no_grad():
            net.denoise_fn(x6.cuda(), lvl.cuda(), guide.cuda())
        torch.cuda.synchronize()
        force = {}
        for Ld in unet_layers(SID):
            key = "denoise_fn." + Ld.name
            force[key] = net.denoise_fn.debug_read(Ld.name, "out").float().cpu()[[0, 15]]
            if Ld.kind == "block":
                force[key + ".res_block.h1"] = net.denoise_fn.debug_read(Ld.name, "h1").float().cpu()[[0, 15]]
        etaps = {}
        O.dy3h_naive_forward_emu(sd, F.pad(x6[[0, 15]], (0, 32, 0, 32), mode="reflect"), lvl[[0, 15]],
                                 F.pad(guide[[0, 15]], (0, 32, 0, 32), mode="reflect"), taps=etaps, force=force)
        out = {k: C.metrics(v, etaps[k].to(torch.bfloat16).float()) for k, v in force.items()}
    else:
        out = C.layerwise_emu_case(SID, B, 256, 256, [0.4, 0.003, 0.8, 0.95][:B], seed=31, net_sd=sid_net)
    worst = max(out, key=lambda k: out[k]["rel_rms"])
    print(f"B = {B}: {len(out)} activations, worst {worst}: {out[worst]}")
    assert len(out) >= 36 + 27                  # 36 layer outputs (stem, 27 blocks, 4 + 4 resamplers) + 27 h1 tensors (+ eps)
    for k, m in out.items():
        assert not m["nan"] and m["rel_rms"] < EMU_LAYER_TOL, (k, m)


@pytest.mark.parametrize("B", [8, 16])
def test_forward_bench_dispatch_vs_oracle_and_reference(golden_dir, sid_net, B):
    """The dispatch bench.py times (B = 16 at 256^2; B = 8 engages the same persistent kernels) against the oracle AND the
    reference's own output (round-3 verdict: every other full-forward parity test runs B <= 2, i.e. the one-shot kernels).
    Sample 0 is the seed-21 input of tests/golden/sid_forward.npz (level t = 25): its eps is compared with the reference's
    crop / subsampled grid; samples 0 and B - 1 are compared with B = 1 oracle forwards.  The profiler's launch keys prove
    that the persistent kernels (akgm_ws 113 / 114 / 115, conv_ws 23, conv_ws128 24, qkv_ws 105)
    ran in this forward."""
    from ucdir_amd.weights import synth_inputs
    net, sd = sid_net
    g = np.load(os.path.join(golden_dir, "sid_forward.npz"))
    c0, g0, x0 = map(torch.from_numpy, synth_inputs(1, 256, 256, seed=21))
    cr, gr, xr = map(torch.from_numpy, synth_inputs(B - 1, 256, 256, seed=77))
    cond, guide, x_t = torch.cat([c0, cr]), torch.cat([g0, gr]), torch.cat([x0, xr])
    lv = [float(g["levels"][1])] + [float(v) for v in np.linspace(0.05, 0.95, B - 1)]
    lvl = torch.tensor(lv, dtype=torch.float32).view(B, 1)
    x6 = torch.cat([cond, x_t], 1)
    L = C.ulib.load()

    def fwd():
        with torch.no_grad():
            e = net.denoise_fn(x6.cuda(), lvl.cuda(), guide.cuda())
        torch.cuda.synchronize()
        return e.cpu()
    eps, keys = _profile_keys(L, fwd)
    want = [113, 114, 115, 116, 23, 24, 105, 127, 128, 129]   # (129: conv_sk_kernel<1,4,9> with wide + short units - the 36^2 level at B = 16, the 72^2 level at B = 8)
    assert all(k in keys for k in want), (sorted(keys), want)
    assert bool(torch.isfinite(eps).all())
    for b in (0, B - 1):
        ref = O.dy3h_forward(sd, x6[b:b + 1], lvl[b:b + 1], guide[b:b + 1])
        m = C.metrics(eps[b:b + 1], ref)
        print(f"B = {B} dispatch, sample {b} vs oracle:", m)
        assert m["rel_rms"] < FWD_TOL, (b, m)
    gm = C.metrics(eps[0, :, 100:132, 60:92], torch.from_numpy(g["eps1_crop"]))
    gd = C.metrics(eps[0, :, ::8, ::8], torch.from_numpy(g["eps1_ds"]))
    print(f"B = {B} dispatch, sample 0 vs reference golden: crop", gm, "subsampled", gd)
    assert gm["rel_rms"] < CROP_TOL and gd["rel_rms"] < CROP_TOL, (gm, gd)


@pytest.mark.parametrize("args", [
    (2, 288, 288, 128, 64, 1, 0, 0),      # cgemm<64> 1x1 at the 288^2 level (the launch hipcc's packed-f32 code got wrong)
    (4, 288, 288, 64, 64, 3, 1, 0),       # cgemm<64> stride-2 Downsample
    (2, 288, 288, 64, 64, 3, 0, 1),       # conv3x3_halo<64> with the GroupNorm fold
    (4, 288, 288, 64, 64, 3, 0, 1),       # conv_ws (persistent): per-tile fixed-point partials, partition-independent
    (2, 144, 144, 128, 128, 3, 0, 1),     # conv3x3_halo<128>
    (2, 72, 72, 256, 256, 3, 2, 0),       # Upsample parity launches
    (1, 18, 18, 1024, 512, 3, 0, 1),      # split-K: partial tiles summed in a fixed order by conv_splitk_finish_kernel
], ids=["1x1_64", "down_64", "halo_64", "ws_64", "halo_128", "up_256", "splitk"])
def test_output_statistics_exact_and_reproducible(args):
    """The (sum, sum of squares) a launch accumulates with fixed-point atomics equal float64 sums of the output it stored
    (up to the bf16 rounding of that output) and are bit-identical from run to run, at the network's real level sizes."""
    m = C.conv_stats_case(*args)
    assert m["outputs_reproducible"] and m["stats_reproducible"], m
    # the stored output is bf16-rounded AFTER the statistics are taken: the rounding noise averages out as 1/sqrt(elements),
    # so the 18^2 x 512 tensor of the split-K case (166 k elements against >= 1.3 M) gets a wider bound
    assert m["stats_rel"] < (1e-4 if args[1] * args[2] < 72 * 72 else 3e-5), m


def test_restoration_is_bit_reproducible(sid_net):
    """super_resolution (predictor + 4 sampler steps with injected noise) twice on the same inputs: identical bits."""
    net, sd = sid_net
    from ucdir_amd.weights import synth_inputs
    dev = torch.device("cuda")
    net.set_new_noise_schedule(dict(schedule="linear", n_timestep=4, linear_start=1e-6, linear_end=0.4), dev)
    cond = torch.from_numpy(synth_inputs(2, 256, 256, seed=8)[0]).to(dev)
    g = torch.Generator().manual_seed(2)
    noises = [torch.randn(2, 3, 256, 256, generator=g) for _ in range(5)]
    net.noise_source = lambda shape, device, k: noises[k].to(device)
    try:
        with torch.no_grad():
            a = net.super_resolution(cond, False).clone()
            b = net.super_resolution(cond, False)
    finally:
        net.noise_source = None
    assert torch.equal(a, b) and bool(torch.isfinite(a).all())


def test_sampler_8_steps_psnr(sid_net):
    m = C.sampler_case(SID, 64, 64, 8, net_sd=sid_net)
    assert m["psnr_u8"] > 35.0, m               # bf16 bound from SURVEY.md §8c


def test_patch_split_matches_oracle():
    """Inter-step patch split (utils/util.py:108-146) with a small threshold: windows batched on the GPU."""
    from oracle import ucdir_oracle as O
    from ucdir_amd.weights import synth_inputs
    net, sd = C.build_net(SMALL)
    net.denoise_fn.patch_threshold = 0
    net.denoise_fn.patch_skip, net.denoise_fn.patch_padding = 128, 32
    cond, guide, x_t = map(torch.from_numpy, synth_inputs(1, 160, 200, seed=5))
    lvl = torch.tensor([[0.5]])
    x6 = torch.cat([cond, x_t], 1)
    ref = O.dy3h_forward(sd, x6, lvl, guide, patch_threshold=0, skip=128, padding=32)
    with torch.no_grad():
        got = net.denoise_fn(x6.cuda(), lvl.cuda(), guide.cuda())
    m = C.metrics(got, ref)
    assert m["rel_rms"] < FWD_TOL, m


def test_sr_val_entry_point(tmp_path, monkeypatch):
    """`sr.py -p val` plumbing on synthetic PNG pairs (reference: sr.py:505-586)."""
    import yaml
    from PIL import Image
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    rs = np.random.RandomState(0)
    for d in ("lq", "gt"):
        os.makedirs(tmp_path / d)
    for i in range(2):
        gt = rs.randint(0, 255, (72, 88, 3)).astype(np.uint8)
        Image.fromarray(gt).save(tmp_path / "gt" / f"{i:03d}.png")
        Image.fromarray((gt * 0.2).astype(np.uint8)).save(tmp_path / "lq" / f"{i:03d}.png")
    cfg = yaml.safe_load(open(os.path.join(root, "config", "sid.yaml")))
    cfg["datasets"]["val"]["data_args"]["dataroot"] = {"lq": str(tmp_path / "lq"), "gt": str(tmp_path / "gt")}
    cfg["model"]["unet"].update(channel_mults=[1, 2, 4], res_blocks=1, attn_res=[32])
    yaml.safe_dump(cfg, open(tmp_path / "sid_small.yaml", "w"))
    monkeypatch.chdir(tmp_path)
    import importlib.util
    spec = importlib.util.spec_from_file_location("sr_entry", os.path.join(root, "sr.py"))
    sr = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(sr)
    psnr, ssim = sr.main(["-p", "val", "-c", str(tmp_path / "sid_small.yaml"), "--synthetic-weights"])
    assert np.isfinite(psnr) and -1.0 <= ssim <= 1.0      # random-noise targets: SSIM ~ 0
    outs = [f for _, _, fs in os.walk(tmp_path / "experiments") for f in fs if f.endswith("_sr.jpg")]
    assert len(outs) == 2


def test_ddim_sample_matches_oracle():
    """5-step strided sampler (model/diffusion.py:247-294) through the same denoiser boundary."""
    from oracle import ucdir_oracle as O
    from ucdir_amd.weights import synth_inputs
    net, sd = C.build_net(SMALL)
    sched = dict(schedule="linear", n_timestep=50, linear_start=1e-6, linear_end=0.4)
    tab = O.schedule_tables(sched)
    net.set_new_noise_schedule(sched, torch.device("cuda"))
    cond, guide, _ = map(torch.from_numpy, synth_inputs(1, 64, 64, seed=9))
    g = torch.Generator().manual_seed(3)
    noises = [torch.randn(1, 3, 64, 64, generator=g) for _ in range(6)]
    ref = O.ddim_sample(sd, tab, cond, guide, noises)
    net.noise_source = lambda shape, device, k: noises[k].to(device)
    with torch.no_grad():
        got = net.ddim_sample(cond.cuda(), kwargs={"guide": guide.cuda()})
    net.noise_source = None
    m = C.metrics(got, ref)
    assert m["rel_rms"] < 3e-2, m


def test_dpm_solver_sample_matches_oracle():
    """DPM-Solver++ multistep sampler (the reference's sr.py:185-231 driver, 20 UNet calls instead of 50) through the
    same denoiser boundary: HIP path vs the oracle's independent restatement with the CPU denoiser."""
    from oracle import ucdir_oracle as O
    from ucdir_amd.weights import synth_inputs
    net, sd = C.build_net(SMALL)
    sched = dict(schedule="linear", n_timestep=50, linear_start=1e-6, linear_end=0.4)
    tab = O.schedule_tables(sched)
    net.set_new_noise_schedule(sched, torch.device("cuda"))
    cond, guide, _ = map(torch.from_numpy, synth_inputs(1, 64, 64, seed=11))
    x_T = torch.randn(1, 3, 64, 64, generator=torch.Generator().manual_seed(5))
    ref = O.dpm_solver_pp_sample(sd, tab, cond, guide, x_T, steps=6, order=2)
    net.noise_source = lambda shape, device, k: x_T.to(device)
    with torch.no_grad():
        got = net.dpm_solver_sample(cond.cuda(), steps=6, order=2, kwargs={"guide": guide.cuda()})
    net.noise_source = None
    m = C.metrics(got, ref)
    assert not m["nan"] and m["rel_rms"] < 3e-2, m


def test_checkpoint_roundtrip(tmp_path):
    """Reference-style EMA checkpoint (`{prefix}_gen_ema.pth`, training-length schedule buffers included,
    model/model.py:193-251) loads through DDPM.load_network and reproduces the forward."""
    from ucdir_amd import model as M
    from ucdir_amd.weights import synth_inputs, synth_state_dict
    import bench
    opt = bench.sid_opt()
    opt["model"]["unet"].update(channel_mults=[1, 2, 4], res_blocks=1, attn_res=[32])
    opt["model"]["beta_schedule"] = {"train": dict(schedule="linear", n_timestep=2000, linear_start=1e-6, linear_end=1e-2),
                                     "val": dict(schedule="linear", n_timestep=50, linear_start=1e-6, linear_end=0.4)}
    opt["train"] = {"ema_scheduler": {"use": True}}
    opt["phase"] = "val"
    net, _ = C.build_net(SMALL, seed=7)
    net.set_new_noise_schedule(opt["model"]["beta_schedule"]["train"], torch.device("cuda"))
    prefix = str(tmp_path / "I100_E1")
    torch.save({k: v.cpu() for k, v in net.state_dict().items()}, prefix + "_gen_ema.pth")
    opt["path"] = {"resume_state": prefix}
    ddpm = M.DDPM(opt)
    ddpm.set_new_noise_schedule(opt["model"]["beta_schedule"]["val"], schedule_phase="val")
    assert ddpm.netG.betas.shape[0] == 50
    cond, guide, x_t = map(torch.from_numpy, synth_inputs(1, 64, 64, seed=2))
    lvl = torch.tensor([[0.4]])
    with torch.no_grad():
        a = net.denoise_fn(torch.cat([cond, x_t], 1).cuda(), lvl.cuda(), guide.cuda())
        b = ddpm.netG.denoise_fn(torch.cat([cond, x_t], 1).cuda(), lvl.cuda(), guide.cuda())
        pa, pb = net.predictor(cond.cuda()), ddpm.netG.predictor(cond.cuda())
    assert torch.equal(a, b) and torch.equal(pa, pb)


@pytest.mark.gpu
def test_alternative_kernel_paths_agree():
    """The default dispatch (resident-weight AKGM, LDS-resident modulation weights, res_conv fused as a 10th tap or as tail
    workgroups of conv1's launch, fused final conv, split-K on small grids, flash attention) and the plain kernels behind the UCDIR_NO_* switches compute the same forward: both are run
    against the oracle in fresh processes (the switches are read once per process)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = {}
    for tag, env_extra in (("default", {}), ("plain", {"UCDIR_NO_PRE": "1", "UCDIR_NO_ATTLDS": "1", "UCDIR_NO_FUSED_RES": "1",
                                                       "UCDIR_NO_FUSED_FINAL": "1", "UCDIR_NO_TAIL_RES": "1", "UCDIR_SPLITK": "0",
                                                       "UCDIR_NO_FLASH": "1"})):
        env = dict(os.environ); env.update(env_extra)
        r = subprocess.run([sys.executable, os.path.join(root, "tools", "gpu_check.py"), "small"], env=env, capture_output=True,
                           text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        line = [l for l in r.stdout.splitlines() if l.startswith("[forward SMALL")][0]
        assert "FAILED" not in line, line
        out[tag] = json.loads(line[line.index("{"):])
    for tag, m in out.items():
        assert not m["eps"]["nan"] and m["eps"]["rel_rms"] < FWD_TOL, (tag, m["eps"])
    assert abs(out["default"]["eps"]["rel_rms"] - out["plain"]["eps"]["rel_rms"]) < 5e-3, (out["default"]["eps"], out["plain"]["eps"])


@pytest.mark.gpu
def test_persistent_kernel_switches_agree_at_a_size_where_they_engage():
    """The round-3 / round-4 switches (UCDIR_NO_WS / NO_WS16 / NO_WS32 / NO_CONV_WS / NO_CONV_WS128 / NO_QKV_WS / NO_ATILE /
    NO_CONV_SK) only change the dispatch from four tiles per CU on: the full SID network at B = 4, 256^2 (tools/gpu_check.py
    sidb4: sample 0 and 3 against B = 1 oracle forwards) with everything on and with everything off, in fresh processes."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    off = {k: "1" for k in ("UCDIR_NO_WS", "UCDIR_NO_WS16", "UCDIR_NO_WS32", "UCDIR_NO_WS64", "UCDIR_NO_CONV_WS", "UCDIR_NO_CONV_WS128",
                            "UCDIR_NO_QKV_WS", "UCDIR_NO_ATILE", "UCDIR_NO_CONV_SK")}
    out = {}
    # ("akgm_tc": the persistent AKGM tails with their fold constants from akgm_tc_kernel launches instead of formed in-kernel, round 5)
    for tag, env_extra in (("default", {}), ("oneshot", off), ("akgm_tc", {"UCDIR_NO_OWNTC": "1"})):
        env = dict(os.environ); env.update(env_extra)
        r = subprocess.run([sys.executable, os.path.join(root, "tools", "gpu_check.py"), "sidb4"], env=env, capture_output=True,
                           text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        line = [l for l in r.stdout.splitlines() if l.startswith("[forward SID B=4")][0]
        assert "FAILED" not in line, line
        out[tag] = json.loads(line[line.index("{"):])
    for tag, m in out.items():
        for b in ("s0", "s3"):
            assert not m[b]["nan"] and m[b]["rel_rms"] < FWD_TOL, (tag, b, m[b])
    assert 113 in out["default"]["keys"] and 23 in out["default"]["keys"] and 24 in out["default"]["keys"], out["default"]["keys"]
    assert out["akgm_tc"]["keys"] == out["default"]["keys"]              # same kernels; the same arithmetic on the same table entries
    for b in ("s0", "s3"):
        assert abs(out["akgm_tc"][b]["rel_rms"] - out["default"][b]["rel_rms"]) < 1e-4 * out["default"][b]["rel_rms"], (out["akgm_tc"][b], out["default"][b])
    assert not any(k in out["oneshot"]["keys"] for k in (113, 114, 115, 116, 23, 24, 105, 125)), out["oneshot"]["keys"]


def test_matrix_rate_probe_is_plausible():
    """ucdir_matrix_rate (bench.py's `roofline.sustained_peak`): an MFMA-only kernel cannot beat the dense bf16 peak of the
    guide, and on conv-like operands the chip holds a lower clock than on small integers."""
    L = C.ulib.load()
    vals = []
    for rnd in (1, 0):
        v = ctypes.c_double(0.0)
        C.ulib.check(L.ucdir_matrix_rate(4000, rnd, ctypes.byref(v), C._st()))
        vals.append(v.value)
    assert 800.0 < vals[0] <= vals[1] * 1.02 and vals[1] < 2600.0, vals
    v = ctypes.c_double(0.0)
    assert L.ucdir_matrix_rate(0, 1, ctypes.byref(v), C._st()) != 0        # bad argument: an error code, not a launch


def test_gather_windows_matches_reflect_pad_and_slices():
    """ucdir_gather_windows (one launch per engine call of the patch split) = F.pad(mode='reflect') + one slice per window
    (utils/util.py:113-137), bit for bit, on a ragged canvas whose last windows are pulled back inside."""
    import torch.nn.functional as F
    from ucdir_amd import patch as P
    from ucdir_amd.ucdir import gather_windows
    g = C.rng(11)
    x = torch.randn(2, 6, 150, 217, generator=g).cuda()
    skip, padding = 96, 16
    pd = P.patch_pad(150, 217, skip, padding)
    xp = F.pad(x, (pd, pd, pd, pd), mode="reflect")
    wins = P.patch_windows(xp.shape[-2], xp.shape[-1], skip, padding)
    assert len(wins) >= 6
    ref = torch.cat([xp[..., a:b, c:d] for (a, b, c, d) in wins], dim=0)
    wd = torch.tensor([[a, c] for (a, b, c, d) in wins], dtype=torch.int32, device="cuda")
    out = gather_windows(x, pd, wd, skip)
    assert out.shape == ref.shape and torch.equal(out, ref)
    # the pad of a canvas smaller than one window (pd > padding): reflect indexing far into the image
    x2 = torch.randn(1, 6, 70, 90, generator=g).cuda()
    pd2 = P.patch_pad(70, 90, skip, padding)
    xp2 = F.pad(x2, (pd2, pd2, pd2, pd2), mode="reflect")
    wins2 = P.patch_windows(xp2.shape[-2], xp2.shape[-1], skip, padding)
    ref2 = torch.cat([xp2[..., a:b, c:d] for (a, b, c, d) in wins2], dim=0)
    out2 = gather_windows(x2, pd2, torch.tensor([[a, c] for (a, b, c, d) in wins2], dtype=torch.int32, device="cuda"), skip)
    assert torch.equal(out2, ref2)
