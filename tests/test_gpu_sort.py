"""GPU parity of the radix sort against the contract of GPURSSorter (gpu_rs.rs:865-884): ascending,
stable, (u32, u32) pairs, count optionally read from device memory.  Bit-exact."""
import os

import numpy as np
import pytest

from variants import env_param, exp_param  # noqa: F401

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=["generic", "depth", "depth9", "depth9k8", exp_param("depth:onesweep", id="depth_fat_onesweep"),
                                        exp_param("depth:coop", id="depth_fat_coop")],
                ids=lambda p: {"generic": "reduce_scan", "depth": "depth_scan_companion", "depth9": "depth_9bit_digits",
                               "depth9k8": "depth_9bit_digits_2048_tiles"}.get(p))
def sort_ctx(ws, request):
    """Four paths to the same contract: the generic sorter (per-tile histograms -> column scan -> scatter), and the
    renderer's depth sorts behind ws_sorter_sort_depth -- the generic sorter carrying a companion value, and the fat-tile
    one-sweep (round 4) as per-pass launches and as ONE launch with device-wide barriers (WS_DEPTH_SORT selects; inputs
    beyond the fat form's 2 M pairs take the generic sorter)."""
    depth = request.param != "generic"
    env = {"WS_DEPTH_DIGIT_BITS": "8", "WS_DEPTH_TILE_KPT": "0"}
    if depth and ":" in request.param:
        env["WS_DEPTH_SORT"] = request.param.split(":")[1]
    if request.param.startswith("depth9"):   # round 6: 9-bit digits (k_dsort9_*), at both tile sizes
        env["WS_DEPTH_DIGIT_BITS"] = "9"
        env["WS_DEPTH_TILE_KPT"] = "8" if request.param.endswith("k8") else "4"
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    c = ws.Context(0)
    c.depth_mode = depth
    for k, v in old.items():
        if v is None:
            del os.environ[k]
        else:
            os.environ[k] = v
    yield c
    c.close()


def test_sort_known_answer(sort_ctx):
    """The reference's own start-up self test: 8192 reversed f32 keys (gpu_rs.rs:295-331)."""
    assert sort_ctx.sort_selftest()


def _check(ws, ctx, oracle, keys, count=None):
    n = len(keys)
    sorter = ws.GPURSSorter(ctx, max(n, 1))
    try:
        if getattr(ctx, "depth_mode", False):  # a companion value rides along: must arrive with its pair
            aux_in = (np.arange(n, dtype=np.uint32) * np.uint32(2654435761)) ^ np.uint32(0x5BD1E995)
            k, p, ax = sorter.sort_host(keys, np.arange(n, dtype=np.uint32), count=count, depth=True, aux=aux_in)
            mm = n if count is None else min(count, n)
            assert np.array_equal(ax[:mm], aux_in[p[:mm]]), "companion values separated from their pairs"
            assert np.array_equal(ax[mm:], aux_in[mm:])
        else:
            k, p = sorter.sort_host(keys, np.arange(n, dtype=np.uint32), count=count)
    finally:
        sorter.close()
    m = n if count is None else min(count, n)
    ok, op = oracle.sort_pairs(keys[:m], np.arange(m, dtype=np.uint32))
    assert np.array_equal(k[:m], ok), "keys differ from the stable reference sort"
    assert np.array_equal(p[:m], op), "payload differs (stability or permutation broken)"
    if m < n:  # elements past the device-side count are never touched
        assert np.array_equal(k[m:], keys[m:])
        assert np.array_equal(p[m:], np.arange(m, n, dtype=np.uint32))


SIZES = [1, 2, 63, 64, 65, 255, 256, 1023, 1024, 1025, 3840, 4095, 4096, 4097, 8192, 12289, 100_000, 131_072, 131_073, 524_288,
         524_289, 1_000_003, 1_048_576, 1_048_577, 2_097_152, 2_097_153]


@pytest.mark.parametrize("n", SIZES)
def test_sort_random_u32(ws, sort_ctx, oracle, n):
    rng = np.random.default_rng(n)
    _check(ws, sort_ctx, oracle, rng.integers(0, 1 << 32, size=n, dtype=np.uint64).astype(np.uint32))


@pytest.mark.parametrize("kind", ["all_equal", "two_values", "few_distinct", "sorted", "reversed", "all_ones",
                                  "depth_like", "low_byte_only", "high_byte_only"])
def test_sort_distributions(ws, sort_ctx, oracle, kind):
    n = 200_003
    rng = np.random.default_rng(7)
    if kind == "all_equal":
        keys = np.full(n, 0x3F800000, dtype=np.uint32)
    elif kind == "two_values":
        keys = rng.integers(0, 2, size=n).astype(np.uint32) * 0x01000000
    elif kind == "few_distinct":
        keys = rng.integers(0, 17, size=n).astype(np.uint32) * 0x00010203
    elif kind == "sorted":
        keys = np.sort(rng.integers(0, 1 << 32, size=n, dtype=np.uint64).astype(np.uint32))
    elif kind == "reversed":
        keys = np.sort(rng.integers(0, 1 << 32, size=n, dtype=np.uint64).astype(np.uint32))[::-1].copy()
    elif kind == "all_ones":
        keys = np.full(n, 0xFFFFFFFF, dtype=np.uint32)  # same bit pattern as the padding keys
    elif kind == "depth_like":
        keys = (rng.uniform(0.0, 37.5, size=n).astype(np.float32)).view(np.uint32)  # bits of zfar - z
    elif kind == "low_byte_only":
        keys = rng.integers(0, 256, size=n).astype(np.uint32)
    else:
        keys = rng.integers(0, 256, size=n).astype(np.uint32) << 24
    _check(ws, sort_ctx, oracle, keys)


@pytest.mark.parametrize("n,count", [(10_000, 0), (10_000, 1), (10_000, 4096), (10_000, 4097), (10_000, 9_999),
                                     (10_000, 10_000), (10_000, 50_000)])
def test_sort_device_side_count(ws, sort_ctx, oracle, n, count):
    """record_sort_indirect: the number of keys lives in device memory (gpu_rs.rs:875-884)."""
    rng = np.random.default_rng(count)
    _check(ws, sort_ctx, oracle, rng.integers(0, 1 << 32, size=n, dtype=np.uint64).astype(np.uint32), count=count)


def test_sort_large_sortedness(ws, sort_ctx):
    """Full-size property check (C3: 5 M keys): sortedness, permutation, stability -- no oracle involved."""
    n = 5_000_000
    rng = np.random.default_rng(99)
    keys = rng.uniform(0.0, 20.0, size=n).astype(np.float32).view(np.uint32)
    keys[::7] = keys[0]  # plenty of duplicates to exercise stability
    sorter = ws.GPURSSorter(sort_ctx, n)
    try:
        k, p = sorter.sort_host(keys, np.arange(n, dtype=np.uint32), depth=getattr(sort_ctx, "depth_mode", False))
    finally:
        sorter.close()
    assert np.all(k[1:] >= k[:-1])
    assert np.array_equal(keys[p], k)
    seen = np.zeros(n, dtype=bool)
    seen[p] = True
    assert seen.all()
    ties = k[1:] == k[:-1]
    assert np.all(p[1:][ties] > p[:-1][ties])


@pytest.mark.parametrize("digit_bits", ["8", "9"])
@pytest.mark.parametrize("nbits", [0, 1, 5, 12, 13, 20, 26, 27, 28, 30, 31, 32])
def test_depth_sort_key_ranges(ws, oracle, nbits, digit_bits, monkeypatch):
    """Keys confined to a range of nbits bits that starts anywhere (a frame's depth keys are: bits of zfar - z), from a
    single value to the full 32 bits: passes whose digit is the same for every key are the degenerate case of every
    histogram and scan in the sorter."""
    n = 300_001
    rng = np.random.default_rng(1000 + nbits)
    span = (1 << nbits) - 1 if nbits else 0
    base = 0 if nbits >= 32 else int(rng.integers(0, (1 << 32) - span))
    keys = (base + rng.integers(0, span + 1, size=n, dtype=np.uint64)).astype(np.uint32)
    if nbits:
        keys[0], keys[1] = np.uint32(base), np.uint32(base + span)   # the range is exactly nbits wide
    monkeypatch.setenv("WS_DEPTH_DIGIT_BITS", digit_bits)
    ctx = ws.Context(0)
    sorter = ws.GPURSSorter(ctx, n)
    try:
        k, p = sorter.sort_host(keys, np.arange(n, dtype=np.uint32), depth=True)
    finally:
        sorter.close()
        ctx.close()
    ok, op = oracle.sort_pairs(keys, np.arange(n, dtype=np.uint32))
    assert np.array_equal(k, ok) and np.array_equal(p, op)
