"""CPU-side tests of the product's host logic: the C ABI loads and exports what
include/bark_hip.h declares, and the forest packer (host C in libbarkhip.so) produces a wire
format whose walk reproduces the reference leaves.  No GPU compute is called here."""
import ctypes
import os
import re

import numpy as np
import pytest

from bark_amd import _lib, forest as bf, synthetic
from oracle import oracle as orc

from conftest import ROOT, load_golden

LEAF, CAT, FEAT = 0x80000000, 0x40000000, 0x3FFFFFFF


def test_library_exports_every_declared_symbol():
    # every header under include/: the drop-in boundary (bark_hip.h) and the test-suite's own hooks (bark_hip_testing.h)
    inc = os.path.join(ROOT, "include")
    header = "".join(open(os.path.join(inc, f)).read() for f in sorted(os.listdir(inc)) if f.endswith(".h"))
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(bark_[a-z0-9_]+)\s*\(", header))
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    lib = _lib.lib()
    for name in declared:
        assert getattr(lib, name) is not None
    assert lib.bark_version() == 210
    assert lib.bark_last_error() == b""
    assert lib.bark_leaf_npad(1) == 128 and lib.bark_leaf_npad(128) == 128 and lib.bark_leaf_npad(129) == 256


def host_pack(nodes3, ft):
    lib = _lib.lib()
    info = _lib.PackInfo()
    B, m, L = nodes3.shape
    nodes3 = np.ascontiguousarray(nodes3)
    ft = np.ascontiguousarray(ft, dtype=np.int64)
    _lib.check(lib.bark_forest_pack_info(_lib.ptr(nodes3), B, m, L, _lib.ptr(ft), ft.shape[0], ctypes.byref(info)))
    packed = np.zeros(info.packed_bytes // 4, dtype=np.uint32)
    _lib.check(lib.bark_forest_pack(_lib.ptr(nodes3), _lib.ptr(ft), ft.shape[0], ctypes.byref(info), _lib.ptr(packed)))
    return info, packed.reshape(B, m, info.stride, 4)


def walk_packed(packed_tree, x, max_depth):
    """numpy emulation of the device walk (traverse.hip) on the wire format."""
    n = packed_tree[0]
    for _ in range(max_depth):
        if n[0] & LEAF:
            break
        f = int(n[0] & FEAT)
        if n[0] & CAT:
            xt = np.trunc(x[f])
            left = bool((int(n[1]) >> int(xt)) & 1) if 0 <= xt < 32 else False
        else:
            left = x[f] <= float(np.uint32(n[1]).view(np.float32))
        n = packed_tree[int(n[2] if left else n[3])]
    assert n[0] & LEAF
    return int(n[1]), int(n[0] & 0xFF), int(n[2])


@pytest.mark.parametrize("name", ["g1_kat_tree", "g3_prior_mixed_n64", "g5_boundaries", "g7_tree_function"])
def test_packer_wire_format_reproduces_reference_leaves(name):
    g = load_golden(name)
    key = "nodes" if "nodes" in g else "forest"
    nodes = orc.nodes_from_raw(g[key])
    nodes3 = nodes.reshape(-1, *nodes.shape[-2:])
    X = g["x"] if "x" in g else g["X"]
    ft = g["feat_types"]
    leaves = g["leaves"].reshape(nodes3.shape[0], X.shape[0], nodes3.shape[1])
    info, packed = host_pack(nodes3, ft)
    assert info.stride <= nodes3.shape[2] and info.max_leaves <= (nodes3.shape[2] + 1) // 2 + 1
    codes = np.zeros((nodes3.shape[0], X.shape[0], (info.max_bits + 31) // 32), dtype=np.uint32)
    for b in range(nodes3.shape[0]):
        for t in range(nodes3.shape[1]):
            dense_of = {}
            for i in range(X.shape[0]):
                orig, dense, bit = walk_packed(packed[b, t], X[i], info.max_depth)
                assert orig == leaves[b, i, t]
                assert dense_of.setdefault(orig, dense) == dense and dense < info.max_leaves
                codes[b, i, bit // 32] |= np.uint32(1) << np.uint32(bit % 32)
                assert bit < info.max_bits
            assert len(set(dense_of.values())) == len(dense_of)  # dense ids are a bijection of reached leaves
    # one-hot code: every point sets exactly one bit per tree, and popcount(z_i & z_j) counts agreeing trees
    pop = np.vectorize(lambda v: bin(int(v)).count("1"))
    for b in range(nodes3.shape[0]):
        assert (pop(codes[b]).sum(axis=1) == nodes3.shape[1]).all()
        agree = (leaves[b][:, None, :] == leaves[b][None, :, :]).sum(-1)
        both = pop(codes[b][:, None, :] & codes[b][None, :, :]).sum(-1)
        assert np.array_equal(agree, both)


def test_packer_statistics_match_active_nodes():
    X, y, bounds, ft = synthetic.mixed_problem(16, 3)
    F = synthetic.sample_prior_forests(4, 50, bounds, ft, seed=11)
    info, packed = host_pack(F, ft)
    active = (F["active"] == 1).sum(-1)
    assert info.stride == active.max()  # prior forests have no pruned garbage: reachable == active
    leaves = ((F["active"] == 1) & (F["is_leaf"] == 1)).sum(-1)
    assert info.max_leaves == leaves.max()
    assert info.max_depth == F["depth"][F["active"] == 1].max()
    assert info.max_bits == leaves.sum(-1).max()
    lib = _lib.lib()
    assert lib.bark_leaf_encoding(ctypes.byref(info)) == 1  # ~3 leaves / tree: the one-hot code is the cheaper one
    assert lib.bark_leaf_words(ctypes.byref(info)) == (info.max_bits + 31) // 32
    deep = synthetic.full_binary_forest(50, 8, 5, np.random.default_rng(0))[None]  # 32 leaves / tree
    dinfo, _ = host_pack(deep, np.full(8, 2))
    assert dinfo.max_bits == 1600 and lib.bark_leaf_encoding(ctypes.byref(dinfo)) == 0
    assert lib.bark_leaf_words(ctypes.byref(dinfo)) == 13


def _tree(rows, L=8):
    t = np.zeros(L, dtype=bf.NODE_RECORD_DTYPE)
    for i, r in enumerate(rows):
        t[i] = r
    return t[None, None]


def test_packer_rejects_malformed_trees():
    ft = np.array([2, 0])
    ok = _tree([(0, 0, 0.5, 1, 2, 0, 0, 1), (1, 0, 0, 0, 0, 0, 1, 1), (1, 0, 0, 0, 0, 0, 1, 1)])
    host_pack(ok, ft)
    with pytest.raises(ValueError, match="outside container"):
        host_pack(_tree([(0, 0, 0.5, 1, 9, 0, 0, 1), (1, 0, 0, 0, 0, 0, 1, 1)]), ft)
    with pytest.raises(ValueError, match="cycle"):
        host_pack(_tree([(0, 0, 0.5, 1, 2, 0, 0, 1), (0, 0, 0.2, 0, 2, 0, 1, 1), (1, 0, 0, 0, 0, 0, 1, 1)]), ft)
    with pytest.raises(ValueError, match="feature_idx"):
        host_pack(_tree([(0, 5, 0.5, 1, 2, 0, 0, 1), (1, 0, 0, 0, 0, 0, 1, 1), (1, 0, 0, 0, 0, 0, 1, 1)]), ft)
    with pytest.raises(ValueError, match="bitmask"):
        host_pack(_tree([(0, 1, -3.0, 1, 2, 0, 0, 1), (1, 0, 0, 0, 0, 0, 1, 1), (1, 0, 0, 0, 0, 0, 1, 1)]), ft)
    # many forests are packed on several host threads: same result as one by one, and the error of a bad
    # forest reaches the caller's message buffer with its index
    many = np.repeat(ok, 96, axis=0)
    info, packed = host_pack(many, ft)
    assert info.B == 96 and np.array_equal(packed[0], packed[95]) and np.array_equal(packed[0], host_pack(ok, ft)[1][0])
    many[70] = _tree([(0, 0, 0.5, 1, 9, 0, 0, 1), (1, 0, 0, 0, 0, 0, 1, 1)])[0]
    with pytest.raises(ValueError, match="forest 70 .*outside container"):
        host_pack(many, ft)
    # a forest that does not fit the info it is packed with is an error, not an overrun
    small_info, _ = host_pack(_tree([(1, 0, 0.5, 0, 0, 0, 0, 1)]), ft)
    buf = np.zeros(small_info.packed_bytes // 4, dtype=np.uint32)
    ft64 = np.ascontiguousarray(ft, dtype=np.int64)
    with pytest.raises(ValueError, match="does not match"):
        _lib.check(_lib.lib().bark_forest_pack(_lib.ptr(np.ascontiguousarray(ok)), _lib.ptr(ft64), 2, ctypes.byref(small_info),
                                               _lib.ptr(buf)))
    # pruned garbage behind a leaf is ignored (tree_proposals.py:168-175 leaves children in place)
    pruned = _tree([(1, 0, 0.5, 7, 7, 0, 0, 1)])
    info, packed = host_pack(pruned, ft)
    assert info.stride == 1 and info.max_leaves == 1 and info.max_depth == 0


def test_workspace_size_is_linear_in_chunk():
    lib = _lib.lib()
    one = lib.bark_mll_workspace_bytes(4096, 0, 50, 1)
    a, b2, many = (lib.bark_mll_workspace_bytes(4096, 0, 50, k) for k in (64, 128, 256))
    assert one >= 4096 * 4096 * 8 and abs((many - b2) - 2 * (b2 - a)) <= 256 * 1024
    sizes = [lib.bark_mll_workspace_bytes(4096, 0, 50, k) for k in range(1, 40)]
    assert all(y > x for x, y in zip(sizes, sizes[1:]))  # monotone (small chunks also carry split-K scratch)
    assert lib.bark_mll_workspace_bytes(0, 0, 50, 1) == 0
    assert lib.bark_mll_workspace_bytes(1000, 500, 50, 3) > lib.bark_mll_workspace_bytes(1000, 0, 50, 3)


def test_reference_api_surface():
    # names a user of bark.forest / tree_gps / quick_inverse / mcmc_record_mll imports
    import bark_amd.fitting as fit
    import bark_amd.tree_kernels as tk
    from bark_amd.tree_kernels.tree_model_kernel import TreeAgreementKernel  # noqa: F401

    for n in ("NODE_RECORD_DTYPE", "FeatureTypeEnum", "create_empty_forest", "_pass_one_through_tree", "pass_through_tree",
              "pass_through_forest", "get_leaf_vectors", "forest_gram_matrix", "batched_forest_gram_matrix",
              "batched_forest_gram_matrix_no_null"):
        assert hasattr(bf, n)
    assert bf.NODE_RECORD_DTYPE == orc.NODE_RECORD_DTYPE and bf.NODE_RECORD_DTYPE.itemsize == 26
    assert [e.value for e in bf.FeatureTypeEnum] == [0, 1, 2]
    assert np.array_equal(bf.create_empty_forest(3, 10), orc.create_empty_forest(3, 10))
    for n in ("mll", "low_rank_inv_update", "low_rank_det_update"):
        assert hasattr(fit.quick_inverse, n)
    assert _lib.lib().bark_lowrank_workspace_bytes(4096, 8) >= 3 * 4096 * 8 * 8
    assert _lib.lib().bark_lowrank_workspace_bytes(4096, 65) == 0
    assert hasattr(fit, "mll") and hasattr(fit, "batched_mll")
    assert hasattr(tk, "forest_predict") and hasattr(tk, "mixture_of_gaussians_as_normal")
    model = tk.BARKModel(bf.create_empty_forest(2, 4)[None], np.array([0.1]), np.array([1.0]))  # tree_gps.py:14-17
    assert model._fields == ("forest", "noise", "scale") and model[1][0] == 0.1
    for n in ("ChainState", "batched_kernel_inverse"):
        assert hasattr(fit, n)
    g = load_golden("g6_predict")
    mix_mu, mix_var = tk.mixture_of_gaussians_as_normal(g["mu"], g["var"])
    assert np.allclose(mix_mu, g["mix_mu"]) and np.allclose(mix_var, g["mix_var"])


def test_product_path_fails_loudly_without_gpu():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    g = load_golden("g1_kat_tree")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        bf.forest_gram_matrix(orc.nodes_from_raw(g["nodes"]), g["x"], g["x"], g["feat_types"])


def test_product_never_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "bark_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in text.replace("no oracle", ""), f"{f} mentions the oracle"


def test_synthetic_prior_forests_are_valid_and_small():
    X, y, bounds, ft = synthetic.mixed_problem(64, 5)
    F = synthetic.sample_prior_forests(3, 50, bounds, ft, seed=42)
    leaves = np.stack([orc.pass_through_forest(F[b], X, ft) for b in range(3)])
    assert leaves.max() < 100
    K = orc.batched_forest_gram_matrix(F, X, X, ft)
    assert np.all(np.diagonal(K, axis1=1, axis2=2) == 1.0) and np.allclose(K, K.transpose(0, 2, 1))
    # deterministic in the seed
    assert np.array_equal(F, synthetic.sample_prior_forests(3, 50, bounds, ft, seed=42))
    deep = synthetic.full_binary_forest(5, 8, 5, np.random.default_rng(0))
    assert ((deep["active"] == 1).sum(-1) == 63).all()


def test_committed_bench_line_follows_the_contract():
    """The last bench line committed under profiles/ carries every field the driver and the judge read."""
    import glob
    import json

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = sorted(glob.glob(os.path.join(root, "profiles", "r*", "bench_c3_final.json")))
    assert files, "no committed bench line"
    r = json.load(open(files[-1]))
    base = json.load(open(os.path.join(root, "BASELINE.json")))
    assert base["metric"].startswith(r["metric"].split(";")[0][:40])
    assert r["unit"] == "evals/s" and r["higher_is_better"] is True and r["scaling"] == "weak"
    assert r["dtype"] == "f64" and r["data"] == "synthetic" and r["vs_baseline"] is None  # BASELINE.md publishes nothing
    assert r["n_gpus"] >= 1 and r["steps"] >= 1 and r["warmup"] >= 0 and r["value"] > 0 and r["ms_per_step"] > 0
    assert "workload" in r["config"] and "model" not in r["config"]
    assert abs(r["value"] - r["config"]["forests_per_gpu"] * r["n_gpus"] / (r["ms_per_step"] * 1e-3)) < 1e-6 * r["value"]
    assert r["config"]["timed_path"].startswith("production")  # timing = NULL: no host sync inside the library
    roof = r["roofline"]
    assert roof["bound"] in ("hbm", "mfma") and roof["unit"] in ("GB/s", "TFLOP/s")
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-12 and 0 < roof["frac"] < 1
    assert roof["traffic"] is None or roof["traffic"] > 0
    # `achieved` is the algorithmic flops over the event-timed call, which cannot be shorter than... nor much longer than a step
    assert 0.9 * r["ms_per_step"] < roof["call_ms"] <= 1.001 * r["ms_per_step"]
    g = roof["gram_kernel"]
    assert g["bound"] == "hbm" and 0 < g["frac"] < 1
    cpu = r["cpu_baseline"]
    assert cpu["kind"] in ("port", "reference") and cpu["cores"] >= 1 and cpu["value"] > 0 and cpu["sample"]
    assert cpu["single_thread"]["cores"] == 1 and cpu["single_thread"]["value"] > 0
    names = " ".join(c["config"] for c in r["configs"])
    assert "c2" in names and "c4" in names and "c5 MLL" in names and "c5 posterior" in names


def test_committed_hbm_profile_names_its_build():
    """roofline.traffic is only quoted from a PMC profile that records the kernel-source digest and workload it was taken on."""
    import glob
    import json

    import bench

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = sorted(glob.glob(os.path.join(root, "profiles", "r02", "*hbm_counters*.json")))
    assert files
    meta = json.load(open(files[-1]))["meta"]
    assert set(("N", "B", "m", "csrc_digest", "steps_profiled", "sweep_kernels", "command")) <= set(meta)
    assert bench.hbm_traffic_from_profile(1, 2, 3) is None  # another workload: never a stale figure
    t = bench.hbm_traffic_from_profile(meta["N"], meta["B"], meta["m"])
    assert t is None or (t["csrc_digest"] == bench.csrc_digest() and t["bytes_per_step"] > 1e10)


def test_packer_walk_bound_is_the_longest_path_when_a_node_has_two_parents():
    """A node shared by two parents is not a tree, but the reference's walk (forest.py:28-47) just follows the child
    pointers.  The packer's max_depth bounds the device walk, so it must be the longest root-to-leaf path even when
    the shared node is first discovered on a shallower one."""
    import ctypes

    from bark_amd import _lib

    nodes = bf.create_empty_forest(1, 8)
    # root(0): left -> 1 (shallow path to the shared node 3), right -> 2 -> 4 -> 3 (deep path);  3: split -> leaves 5, 6
    nodes[0, 0] = (0, 0, 0.5, 1, 2, 0xFFFFFFFF, 0, 1)
    nodes[0, 1] = (0, 0, 0.25, 3, 7, 0, 1, 1)
    nodes[0, 2] = (0, 0, 0.75, 4, 7, 0, 1, 1)
    nodes[0, 4] = (0, 0, 0.6, 3, 7, 2, 2, 1)
    nodes[0, 3] = (0, 0, 0.1, 5, 6, 1, 2, 1)
    for leaf in (5, 6, 7):
        nodes[0, leaf] = (1, 0, 0, 0, 0, 0, 3, 1)
    ft = np.array([2], dtype=np.int64)
    info = _lib.PackInfo()
    _lib.check(_lib.lib().bark_forest_pack_info(_lib.ptr(nodes), 1, 1, 8, _lib.ptr(ft), 1, ctypes.byref(info)))
    assert info.max_depth == 4  # 0 -> 2 -> 4 -> 3 -> leaf
    # the oracle follows the pointers like the reference: x = 0.55 takes the deep path and ends on leaf 6
    assert orc.pass_through_forest(nodes, np.array([[0.55], [0.2], [0.05]]), ft)[:, 0].tolist() == [6, 6, 5]


def test_context_api_without_a_gpu_fails_with_a_status_not_a_crash():
    """bark_ctx_* are plain C calls: on a host without a GPU creation reports an error code and message; destroy(NULL)
    and the size query on NULL are harmless.  (On a GPU box creation succeeds and the context is destroyed again.)"""
    import ctypes

    from bark_amd import _lib

    lib = _lib.lib()
    h = ctypes.c_void_p()
    rc = lib.bark_ctx_create(0, ctypes.byref(h))
    if rc == 0:
        assert h.value
        lib.bark_ctx_destroy(h)
    else:
        assert not h.value and lib.bark_last_error() != b""
    assert lib.bark_ctx_create(0, None) != 0
    lib.bark_ctx_destroy(None)
    assert lib.bark_ctx_workspace_bytes(None) == 0
    # entry points refuse a null context before touching anything else
    assert lib.bark_leaf_indices_hip(None, None, None, None, 1, 1, None, None) == _lib.BARK_ERR_ARG
    assert b"bark_ctx" in lib.bark_last_error()
    assert lib.bark_mll_batched_hip(None, None, None, None, 1, 1, None, None, None, None, 0, None, 0, None, None, None, None, None,
                                    None, 0, 1, None, None) == _lib.BARK_ERR_ARG


def test_workgroup_to_tile_map_reaches_every_pair_once():
    """The sweep kernels share one workgroup -> (matrix, tile) map (XCD-aware placement, chol.hip xcd_map): chunks smaller than 8
    matrices, and small chunks whose size is not a multiple of 8, are dealt out as "virtual matrices" so that no XCD carries
    twice the work of another.  Placement is speed only — but the map has to be a bijection onto the launch's pairs."""
    from bark_amd import _lib

    lib = _lib.lib()
    for Bc in list(range(1, 81)) + [96, 100, 250, 256, 1000]:
        for ntiles in (1, 2, 3, 7, 8, 15, 31, 32, 100, 248):
            assert lib.bark_xcd_map_selftest(ntiles, Bc) == 0, (ntiles, Bc)
    assert lib.bark_xcd_map_selftest(0, 4) == -1


# DESIGN.md section 4 "Which schedule runs": (N, B, C, chunk) -> (schedule, fused Gram, device-side wait, gates, diag_pre_kernel)
SCHEDULE_TABLE = {
    "c1 N=64 x 1": ((64, 1, 0, None), ("one_block", 1, 0, 0, 0)),
    "c2 N=1024 x 1": ((1024, 1, 0, None), ("splitk", 0, 1, 0, 1)),
    "c3 N=4096 x 256": ((4096, 256, 0, None), ("paired", 1, 0, 0, 0)),
    "c4 share N=4096 x 64": ((4096, 64, 0, None), ("pipelined", 1, 0, 0, 0)),
    "c4 on one GPU, N=4096 x 512 in chunks of 256": ((4096, 512, 0, 256), ("paired", 1, 0, 0, 0)),
    "c5 N=16384 x 1": ((16384, 1, 0, None), ("pipelined", 1, 1, 1, 0)),
    "c5 posterior, 10^4 candidates": ((16384, 1, 10000, None), ("pipelined", 0, 1, 1, 0)),
    "c5 B=4 variant": ((16384, 4, 0, None), ("pipelined", 1, 1, 1, 0)),
    "lone N=4096": ((4096, 1, 0, None), ("splitk", 0, 1, 0, 1)),
    "N=4096 x 8": ((4096, 8, 0, None), ("pipelined", 1, 1, 1, 0)),
    "N=4096 x 16": ((4096, 16, 0, None), ("pipelined", 1, 1, 1, 0)),
    "N=2048 x 256 (16 block rows)": ((2048, 256, 0, None), ("paired", 1, 0, 0, 0)),
    "N=1920 x 256 (15 block rows)": ((1920, 256, 0, None), ("pipelined", 1, 0, 0, 0)),
    "N=512 x 256 (four block rows: one launch)": ((512, 256, 0, None), ("multi_block", 1, 0, 0, 0)),
    "N=512 x 64": ((512, 64, 0, None), ("splitk", 0, 0, 0, 0)),
    "N=256 x 256 (two block rows: one launch)": ((256, 256, 0, None), ("two_block", 1, 0, 0, 0)),
    "N=384 x 256 (three block rows: one launch)": ((384, 256, 0, None), ("multi_block", 1, 0, 0, 0)),
    "N=384 x 512 (two full rounds of the CUs)": ((384, 512, 0, None), ("multi_block", 1, 0, 0, 0)),
    "N=512 x 300 (a second round that would be mostly empty: the sweep)": ((512, 300, 0, None), ("plain", 1, 0, 0, 0)),
    "N=768 x 256 (six block rows: one launch)": ((768, 256, 0, None), ("multi_block", 1, 0, 0, 0)),
    "N=768 x 384 (1.5 rounds of the CUs: the sweep)": ((768, 384, 0, None), ("plain", 1, 0, 0, 0)),
    "N=384 x 64 (below the window: the sweep)": ((384, 64, 0, None), ("plain", 1, 0, 0, 0)),
    "N=64 x 256": ((64, 256, 0, None), ("one_block", 1, 0, 0, 0)),
    "lone N=6900 (look-ahead)": ((6900, 1, 0, None), ("splitk_lookahead", 0, 1, 1, 1)),
}


def test_schedule_table_of_the_baseline_configs():
    """which of Sweep's schedules each BASELINE config (and each `configs` row of the bench line) takes, asked
    through the ABI (bark_mll_plan_query = the function bark_mll_batched_hip configures its sweep from) — a changed tuning
    constant that moves a shape, above all the headline c3 off the paired schedule, fails here.  No GPU needed."""
    from bark_amd.fitting import schedule_plan

    prev = _lib.lib().bark_device_wait(1)  # the table is for the default (mechanism on)
    try:
        for name, ((N, B, C, chunk), want) in SCHEDULE_TABLE.items():
            d = schedule_plan(N, B, C=C, chunk=chunk)
            got = (d["schedule"], d["fused_gram"], d["dev_wait"], d["dev_gate"], d["pre_update"])
            assert got == want, (name, got, want)
            assert d["last_schedule"] == d["schedule"], name
        d = schedule_plan(4096, 300, chunk=256)  # a ragged last chunk takes its own schedule
        assert (d["n_chunks"], d["last_chunk"], d["schedule"], d["last_schedule"]) == (2, 44, "paired", "pipelined")
        assert schedule_plan(64, 1, timing=True)["schedule"] == "plain"  # the instrumented call has no one-launch form
        # two block rows in one launch (round 5): chunks of up to 384 matrices, larger ones up to N = 224; otherwise the sweep
        assert [schedule_plan(256, b)["schedule"] for b in (1, 16, 384, 385)] == ["two_block", "two_block", "two_block", "plain"]
        assert schedule_plan(224, 2048)["schedule"] == "two_block" and schedule_plan(225, 2048)["schedule"] == "plain"
        _lib.lib().bark_device_wait(0)
        assert schedule_plan(1024, 1)["dev_wait"] == 0 and schedule_plan(4096, 8)["dev_gate"] == 0
    finally:
        _lib.lib().bark_device_wait(prev)
