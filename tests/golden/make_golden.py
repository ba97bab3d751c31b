"""Generates the committed golden fixtures under tests/golden/ (run once, in the build container).

Two kinds of fixture, kept apart by file name:

* ``ref_*.npz``  -- inputs + outputs of the REFERENCE's own importable Python, loaded by file path from
  /root/reference (read-only; the reference cannot travel to the GPU box, so only these vectors do):
    - vision3d/ops/pairwise_distance.py         (pure torch, CPU)          -> ref_pairwise_distance.npz
    - vision3d/array_ops/knn.py                 (scipy cKDTree.query)       -> ref_array_knn.npz
    - vision3d/array_ops/ball_query.py          (cKDTree.query_ball_point)  -> ref_array_ball_query.npz
    - vision3d/array_ops/furthest_point_sample.py  (native_furthest_point_sample, NumPy) -> ref_array_fps.npz
    - vision3d/ops/knn_interpolate.py, ops/point_cloud_partition.py (pure torch once `.knn` resolves) -> ref_knn_ops.npz
    - vision3d/ops/grid_subsample.py (naive :13-46 as is, fast :49-99), ops/conversion.py:7-83, and the pure-torch glue of
      ops/radius_search.py:59-70 around a stand-in kNN                                    -> ref_pack_glue.npz
  The CUDA kernels themselves cannot be built or run here (no nvcc, no GPU) and ship no test vectors.
* ``oracle_*.npz`` -- regression vectors of OUR CPU oracle on tie-stress inputs (lattices, duplicates, N<k,
  empty balls).  They pin the oracle against drift and give the GPU tests a file-based target; they are NOT
  evidence about the reference (tie order there is "parity unpinned").

Usage:  python tests/golden/make_golden.py
"""
import importlib.util
import os
import sys
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference/vision3d"
sys.path.insert(0, REPO)

import oracle  # noqa: E402


def load_by_path(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        spec.loader.exec_module(mod)
    return mod


def unit_sphere_cloud(rng, b, n):
    """ModelNet40-shaped synthetic cloud: N(0,1)^3, centred, scaled into the unit sphere."""
    p = rng.standard_normal((b, n, 3)).astype(np.float32)
    p -= p.mean(axis=1, keepdims=True)
    p /= np.linalg.norm(p, axis=2).max(axis=1)[:, None, None]
    return p.astype(np.float32)


def make_reference_fixtures():
    rng = np.random.default_rng(20261110)
    # ---- pairwise_distance: every flag combination of the reference function --------------------------
    pd = load_by_path("ref_pairwise_distance", os.path.join(REF, "ops", "pairwise_distance.py"))
    x = rng.standard_normal((2, 37, 20)).astype(np.float32)
    y = rng.standard_normal((2, 45, 20)).astype(np.float32)
    xn = x / np.linalg.norm(x, axis=-1, keepdims=True)
    yn = y / np.linalg.norm(y, axis=-1, keepdims=True)
    out = {"x": x, "y": y, "xn": xn.astype(np.float32), "yn": yn.astype(np.float32)}
    for normalized in (False, True):
        for squared in (False, True):
            for strict in (False, True):
                for transposed in (False, True):
                    a, b = (xn, yn) if normalized else (x, y)
                    ta, tb = torch.from_numpy(a.astype(np.float32)), torch.from_numpy(b.astype(np.float32))
                    if transposed:
                        ta, tb = ta.transpose(-1, -2).contiguous(), tb.transpose(-1, -2).contiguous()
                    d = pd.pairwise_distance(ta, tb, normalized=normalized, transposed=transposed, squared=squared,
                                             strict=strict)
                    key = f"d_n{int(normalized)}_s{int(squared)}_t{int(strict)}_T{int(transposed)}"
                    out[key] = d.numpy().astype(np.float32)
    np.savez_compressed(os.path.join(HERE, "ref_pairwise_distance.npz"), **out)

    # ---- array_ops.knn / array_ops.ball_query (scipy) -------------------------------------------------
    aknn = load_by_path("ref_array_knn", os.path.join(REF, "array_ops", "knn.py"))
    aball = load_by_path("ref_array_ball_query", os.path.join(REF, "array_ops", "ball_query.py"))
    s = unit_sphere_cloud(rng, 1, 600)[0]
    q = s[rng.permutation(600)[:150]] + rng.normal(0, 0.01, (150, 3)).astype(np.float32)
    q = q.astype(np.float32)
    kd, ki = aknn.knn(q, s, k=8, return_distance=True)
    np.savez_compressed(os.path.join(HERE, "ref_array_knn.npz"), q=q, s=s, k=np.int64(8),
                        dist=kd.astype(np.float64), idx=ki.astype(np.int64))
    radius = 0.2
    lists = aball.ball_query(q, s, radius)
    flat = np.concatenate([np.sort(np.asarray(l, dtype=np.int64)) for l in lists]) if len(lists) else np.zeros(0)
    lens = np.asarray([len(l) for l in lists], dtype=np.int64)
    np.savez_compressed(os.path.join(HERE, "ref_array_ball_query.npz"), q=q, s=s, radius=np.float32(radius),
                        flat=flat.astype(np.int64), lens=lens)


def make_select_fixtures():
    """ops/mutual_topk_select.py, run on the CPU: its `torch.arange(..).cuda()` calls (:34,:39,:93,:98,:105) are the
    only GPU dependence, so `Tensor.cuda` is made the identity while the module runs.  Scores are tie-free."""
    orig_cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        mts = load_by_path("ref_mutual_topk_select", os.path.join(REF, "ops", "mutual_topk_select.py"))
        rng = np.random.default_rng(20261111)
        out = {}
        cases = {"a": (1, 37, 53, 3), "b": (3, 64, 40, 1), "c": (2, 130, 200, 5)}
        for name, (b, n, m, k) in cases.items():
            score = rng.standard_normal((b, n, m)).astype(np.float32)
            rmask = rng.random((b, n)) > 0.2
            cmask = rng.random((b, m)) > 0.2
            out[f"{name}_score"], out[f"{name}_k"] = score, np.int64(k)
            out[f"{name}_rmask"], out[f"{name}_cmask"] = rmask, cmask
            ts = torch.from_numpy(score)
            for largest in (True, False):
                for mutual in (True, False):
                    for thr in (None, 0.25 if largest else -0.25):
                        tag = f"{name}_l{int(largest)}_m{int(mutual)}_t{int(thr is not None)}"
                        corr = mts.batch_mutual_topk_select(ts, k, largest=largest, threshold=thr, mutual=mutual,
                                                            reduce_result=False)
                        out[tag] = corr.numpy()
                        corr = mts.batch_mutual_topk_select(ts, k, row_masks=torch.from_numpy(rmask),
                                                            col_masks=torch.from_numpy(cmask), largest=largest,
                                                            threshold=thr, mutual=mutual, reduce_result=False)
                        out[tag + "_masked"] = corr.numpy()
            bi, ri, ci, sc = mts.batch_mutual_topk_select(ts, k)
            out[f"{name}_reduced"] = np.stack([bi.numpy(), ri.numpy(), ci.numpy()])
            out[f"{name}_reduced_scores"] = sc.numpy()
        # ops/matching.py (max selection on log-scores): loaded as a submodule of an empty synthetic parent package whose
        # __path__ is the reference's ops directory, so that its relative imports resolve to the reference's own files
        import importlib
        import types

        parent = types.ModuleType("v3dref_ops")
        parent.__path__ = [os.path.join(REF, "ops")]
        sys.modules["v3dref_ops"] = parent
        try:
            matching = importlib.import_module("v3dref_ops.matching")
            logits = rng.standard_normal((41, 57)).astype(np.float32)
            log_p = logits - np.log(np.exp(logits).sum(axis=1, keepdims=True))
            out["m_log_p"] = log_p.astype(np.float32)
            for mutual in (False, True):
                for dustbin in (False, True):
                    for thr in (0.0, 0.05):
                        r_, c_, s_ = matching.extract_correspondences_from_scores(
                            torch.from_numpy(out["m_log_p"]), mutual=mutual, has_dustbin=dustbin, threshold=thr,
                            return_score=True)
                        tag = f"m_m{int(mutual)}_d{int(dustbin)}_t{int(thr > 0)}"
                        out[tag] = np.stack([r_.numpy(), c_.numpy()])
                        out[tag + "_scores"] = s_.numpy()
        finally:
            for name in [k_ for k_ in sys.modules if k_.startswith("v3dref_ops")]:
                del sys.modules[name]
        ri, ci, sc = mts.mutual_topk_select(torch.from_numpy(out["a_score"][0]), 3, largest=False, mutual=False)
        out["a2d_reduced"] = np.stack([ri.numpy(), ci.numpy()])
        out["a2d_reduced_scores"] = sc.numpy()
        np.savez_compressed(os.path.join(HERE, "ref_select.npz"), **out)
    finally:
        torch.Tensor.cuda = orig_cuda


def make_matching_fixtures():
    """ops/matching.py:64-121 (threshold and global top-k selection) and the two superpoint matching heads
    (models/geotransformer/superpoint_matching_{global,mutual}_topk.py), run on the CPU.  matching.py is loaded as a
    submodule of an empty synthetic parent package (as in make_select_fixtures); the heads import `vision3d.ops`, which is
    given to them as a synthetic module holding the reference's own pairwise_distance / mutual_topk_select loaded by file
    path (the real ops/__init__.py pulls in pykeops and the unbuilt extension).  Scores are tie-free."""
    import importlib
    import types

    orig_cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    parent = types.ModuleType("v3dref_ops")
    parent.__path__ = [os.path.join(REF, "ops")]
    sys.modules["v3dref_ops"] = parent
    saved = {n: sys.modules.get(n) for n in ("vision3d", "vision3d.ops")}
    try:
        matching = importlib.import_module("v3dref_ops.matching")
        rng = np.random.default_rng(20261113)
        out = {}
        logits = rng.standard_normal((45, 61)).astype(np.float32)
        log_p = (logits - np.log(np.exp(logits).sum(axis=1, keepdims=True))).astype(np.float32)
        out["log_p"] = log_p
        t = torch.from_numpy(log_p)
        for i, thr in enumerate((0.02, 0.05, 0.5)):
            r_, c_, s_ = matching.extract_correspondences_from_scores_threshold(t, thr, return_score=True)
            out[f"thr{i}"], out[f"thr{i}_value"] = np.stack([r_.numpy(), c_.numpy()]), np.float32(thr)
            out[f"thr{i}_scores"] = s_.numpy()
        for k in (1, 17, 300):
            for largest in (True, False):
                for dustbin in (False, True):
                    r_, c_, s_ = matching.extract_correspondences_from_scores_topk(
                        t, k, has_dustbin=dustbin, largest=largest, return_score=True)
                    tag = f"topk{k}_l{int(largest)}_d{int(dustbin)}"
                    out[tag], out[tag + "_scores"] = np.stack([r_.numpy(), c_.numpy()]), s_.numpy()

        pd = load_by_path("ref_pairwise_distance", os.path.join(REF, "ops", "pairwise_distance.py"))
        mts = load_by_path("ref_mutual_topk_select", os.path.join(REF, "ops", "mutual_topk_select.py"))
        pkg, ops_mod = types.ModuleType("vision3d"), types.ModuleType("vision3d.ops")
        pkg.__path__ = []
        ops_mod.pairwise_distance, ops_mod.mutual_topk_select = pd.pairwise_distance, mts.mutual_topk_select
        pkg.ops = ops_mod
        sys.modules["vision3d"], sys.modules["vision3d.ops"] = pkg, ops_mod
        geo = os.path.join(REF, "models", "geotransformer")
        g_mod = load_by_path("ref_spm_global", os.path.join(geo, "superpoint_matching_global_topk.py"))
        m_mod = load_by_path("ref_spm_mutual", os.path.join(geo, "superpoint_matching_mutual_topk.py"))
        src = rng.standard_normal((70, 32)).astype(np.float32)
        tgt = rng.standard_normal((83, 32)).astype(np.float32)
        src /= np.linalg.norm(src, axis=1, keepdims=True)
        tgt /= np.linalg.norm(tgt, axis=1, keepdims=True)
        smask, tmask = rng.random(70) > 0.15, rng.random(83) > 0.15
        sw, tw = rng.random(70).astype(np.float32) + 0.5, rng.random(83).astype(np.float32) + 0.5
        out.update(spm_src=src.astype(np.float32), spm_tgt=tgt.astype(np.float32), spm_smask=smask, spm_tmask=tmask,
                   spm_sw=sw, spm_tw=tw)
        ts, tt = torch.from_numpy(out["spm_src"]), torch.from_numpy(out["spm_tgt"])
        for dual in (True, False):
            for masked in (False, True):
                for weighted in (False, True):
                    head = g_mod.SuperPointMatchingGlobalTopk(48, dual_normalization=dual)
                    a, b, c = head(ts, tt,
                                   torch.from_numpy(smask) if masked else None, torch.from_numpy(tmask) if masked else None,
                                   torch.from_numpy(sw) if weighted else None, torch.from_numpy(tw) if weighted else None)
                    tag = f"spmg_d{int(dual)}_m{int(masked)}_w{int(weighted)}"
                    out[tag], out[tag + "_scores"] = np.stack([a.numpy(), b.numpy()]), c.numpy()
        for mutual in (True, False):
            for masked in (False, True):
                for thr in (None, 1.05, 0.9):
                    head = m_mod.SuperPointMatchingMutualTopk(24, 3, threshold=thr, mutual=mutual)
                    a, b, c = head(ts, tt, torch.from_numpy(smask) if masked else None,
                                   torch.from_numpy(tmask) if masked else None)
                    tag = f"spmm_m{int(mutual)}_k{int(masked)}_t{'n' if thr is None else str(thr)}"
                    out[tag], out[tag + "_scores"] = np.stack([a.numpy(), b.numpy()]), c.numpy()
        np.savez_compressed(os.path.join(HERE, "ref_matching.npz"), **out)
    finally:
        torch.Tensor.cuda = orig_cuda
        for name in [k_ for k_ in sys.modules if k_.startswith("v3dref_ops")]:
            del sys.modules[name]
        for n, m in saved.items():
            if m is None:
                sys.modules.pop(n, None)
            else:
                sys.modules[n] = m


def make_voxelize_fixtures():
    """ops/voxelize.py (pure torch) run on the CPU as a submodule of the synthetic parent package (it imports
    .conversion relatively); its `.cuda()` calls on freshly built tensors are the identity while it runs."""
    import importlib
    import types

    orig_cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    parent = types.ModuleType("v3dref_ops")
    parent.__path__ = [os.path.join(REF, "ops")]
    sys.modules["v3dref_ops"] = parent
    try:
        vox = importlib.import_module("v3dref_ops.voxelize")
        rng = np.random.default_rng(20261114)
        lengths = np.asarray([900, 0, 1300, 700], dtype=np.int64)
        pts = (rng.random((int(lengths.sum()), 3)).astype(np.float32) * np.float32([3.0, 3.0, 2.5])
               - np.float32([1.0, 0.5, 0.0]))  # negative coordinates too
        feats = rng.standard_normal((pts.shape[0], 8)).astype(np.float32)
        out = {"points": pts, "lengths": lengths, "feats": feats, "voxel_size": np.float32(0.3)}
        v, n, f, l = vox.voxelize_pack_mode(torch.from_numpy(pts), torch.from_numpy(lengths), 0.3,
                                            feats=torch.from_numpy(feats))
        out.update(voxels=v.numpy(), nodes=n.numpy(), v_feats=f.numpy(), v_lengths=l.numpy())
        v2, n2, l2 = vox.voxelize_pack_mode(torch.from_numpy(pts), torch.from_numpy(lengths), 0.3)
        assert torch.equal(v, v2) and torch.equal(n, n2) and torch.equal(l, l2)
        bpts = rng.random((3, 400, 3)).astype(np.float32)
        bmask = rng.random((3, 400)) > 0.3
        bfeats = rng.standard_normal((3, 400, 4)).astype(np.float32)
        bv, bn, bf, bm = vox.voxelize(torch.from_numpy(bpts), 0.25, masks=torch.from_numpy(bmask),
                                      feats=torch.from_numpy(bfeats))
        out.update(b_points=bpts, b_masks=bmask, b_feats=bfeats, b_voxels=bv.numpy(), b_nodes=bn.numpy(),
                   b_v_feats=bf.numpy(), b_v_masks=bm.numpy())
        np.savez_compressed(os.path.join(HERE, "ref_voxelize.npz"), **out)
    finally:
        torch.Tensor.cuda = orig_cuda
        for name in [k_ for k_ in sys.modules if k_.startswith("v3dref_ops")]:
            del sys.modules[name]


def make_pack_ops_fixtures():
    """ops/index_select.py, ops/pooling.py, ops/nearest_interpolate.py (pure torch) run on the CPU.  They import each other
    relatively, so they are loaded as submodules of an empty synthetic parent package whose __path__ is the reference's ops
    directory (the real ops/__init__.py pulls in pykeops and the unbuilt extension).  ops/knn_interpolate.py imports
    ops/knn.py -> pykeops at module level: its own logic is pinned separately by make_knn_ops_fixtures (ref_knn_ops.npz)."""
    import types

    parent = types.ModuleType("v3dref_ops")
    parent.__path__ = [os.path.join(REF, "ops")]
    sys.modules["v3dref_ops"] = parent
    try:
        import importlib

        pooling = importlib.import_module("v3dref_ops.pooling")
        nearest = importlib.import_module("v3dref_ops.nearest_interpolate")
        isel = importlib.import_module("v3dref_ops.index_select")
        rng = np.random.default_rng(20261112)
        n, m, k, c = 300, 170, 9, 24
        feats = rng.standard_normal((n, c)).astype(np.float32)
        table = rng.integers(0, n + 1, size=(m, k)).astype(np.int64)  # value n = "no neighbour"
        table[5] = n  # a query without any neighbour
        lengths = np.asarray([120, 0 + 80, 100], dtype=np.int64)
        tf, tt = torch.from_numpy(feats), torch.from_numpy(table)
        padded = torch.cat([tf, torch.zeros_like(tf[:1])], 0)
        out = {
            "feats": feats, "table": table, "lengths": lengths,
            "index_select": isel.index_select(padded, tt, dim=0).numpy(),
            "maxpool": pooling.local_maxpool_pack_mode(tf, tt).numpy(),
            "avgpool": pooling.global_avgpool_pack_mode(tf, torch.from_numpy(lengths)).numpy(),
            "nearest": nearest.nearest_interpolate_pack_mode(tf, tt).numpy(),
        }
        # autograd of the reference formulation = the backward the native kernels must reproduce (to rounding)
        tf2 = tf.clone().requires_grad_(True)
        up = torch.from_numpy(rng.standard_normal((m, c)).astype(np.float32))
        pooling.local_maxpool_pack_mode(tf2, tt).backward(up)
        out["maxpool_upstream"], out["maxpool_grad"] = up.numpy(), tf2.grad.numpy()
        tf3 = tf.clone().requires_grad_(True)
        up3 = torch.from_numpy(rng.standard_normal((m, k, c)).astype(np.float32))
        isel.index_select(torch.cat([tf3, torch.zeros_like(tf3[:1])], 0), tt, dim=0).backward(up3)
        out["index_select_upstream"], out["index_select_grad"] = up3.numpy(), tf3.grad.numpy()
        np.savez_compressed(os.path.join(HERE, "ref_pack_ops.npz"), **out)
    finally:
        for name in [k_ for k_ in sys.modules if k_.startswith("v3dref_ops")]:
            del sys.modules[name]


def _synthetic_vision3d(ext_functions=()):
    """Registers an empty `vision3d` package whose `utils.misc` is the reference's own file (loaded by path: it imports
    only torch) and whose `ext` is an empty namespace with the asked-for attribute names, so that modules which call
    `load_ext("vision3d.ext", [...])` at import time (utils/misc.py:23-27) can be loaded without the unbuilt extension.
    Returns the sys.modules entries to restore."""
    import types

    names = ("vision3d", "vision3d.utils", "vision3d.utils.misc", "vision3d.ext")
    saved = {n: sys.modules.get(n) for n in names}
    pkg, utils, ext = types.ModuleType("vision3d"), types.ModuleType("vision3d.utils"), types.ModuleType("vision3d.ext")
    pkg.__path__, utils.__path__ = [], []
    misc = load_by_path("vision3d.utils.misc", os.path.join(REF, "utils", "misc.py"))

    def _absent(*_a, **_k):
        raise RuntimeError("vision3d.ext is not built in this container")

    for f in ext_functions:
        setattr(ext, f, _absent)
    pkg.utils, pkg.ext, utils.misc = utils, ext, misc
    sys.modules.update({"vision3d": pkg, "vision3d.utils": utils, "vision3d.utils.misc": misc, "vision3d.ext": ext})
    return saved


def _restore_modules(saved):
    for n, m in saved.items():
        if m is None:
            sys.modules.pop(n, None)
        else:
            sys.modules[n] = m


def fps_round_gaps(points, indices):
    """Relative gap between the selected (largest) and the runner-up running min-distance of every FPS round, float64."""
    p = points.astype(np.float64)
    dist = np.full(p.shape[0], np.inf)
    gaps = []
    for j in range(len(indices) - 1):
        dist = np.minimum(dist, np.linalg.norm(p - p[indices[j]], axis=1))
        top = np.sort(dist)[-2:]
        gaps.append((top[1] - top[0]) / top[1])
    return np.asarray(gaps)


def make_fps_fixtures():
    """array_ops/furthest_point_sample.py:22-66 `native_furthest_point_sample` (pure NumPy, float64 norms), run by the
    reference's own file.  The file calls load_ext at import time (:6-8), hence the synthetic parent package.  The clouds are
    float32 values handed over as float64 (exact), and only clouds whose every round separates winner and runner-up by more
    than 1e-5 relative are kept (seed bumped otherwise, see `fps_round_gaps`): a float64 reference cannot arbitrate an fp32
    kernel inside fp32 rounding, and among exact ties the CUDA kernel's order is its launch geometry's (unpinnable)."""
    saved = _synthetic_vision3d(ext_functions=("sample_nodes_with_fps",))
    try:
        afps = load_by_path("ref_array_fps", os.path.join(REF, "array_ops", "furthest_point_sample.py"))
        out = {}
        for name, n, m, kind in (("u100", 100, 40, "uniform"), ("u257", 257, 64, "uniform"),
                                 ("s1024", 1024, 128, "sphere"), ("u1000", 1000, 100, "uniform")):
            seed = 20261120
            while True:
                rng = np.random.default_rng(seed)
                pts = rng.random((n, 3), dtype=np.float32) if kind == "uniform" else unit_sphere_cloud(rng, 1, n)[0]
                idx = afps.native_furthest_point_sample(pts.astype(np.float64), min_distance=-1.0, num_samples=m)
                if fps_round_gaps(pts, idx).min() > 1e-5:
                    break
                seed += 1
            out[f"{name}_pts"], out[f"{name}_idx"], out[f"{name}_seed"] = pts, idx.astype(np.int64), np.int64(seed)
            assert idx.shape == (m,) and idx[0] == 0
        np.savez_compressed(os.path.join(HERE, "ref_array_fps.npz"), **out)
    finally:
        _restore_modules(saved)


def make_knn_ops_fixtures():
    """ops/knn_interpolate.py and ops/point_cloud_partition.py, run on the CPU.  Both import `.knn` (pykeops, absent) at
    module level; their own logic is pure torch.  They are loaded as submodules of the synthetic parent package with a
    stand-in `.knn` submodule whose `keops_knn` / `knn` are built on the REFERENCE's own pairwise_distance + torch.topk --
    the in-repo pure-torch equivalent the reference itself uses (models/geotransformer/geometric_structure_embedding.py
    :52,:60).  Inputs are tie-free.  What this pins: everything in the two files except pykeops' arithmetic."""
    import importlib
    import types

    orig_cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    saved = _synthetic_vision3d()
    parent = types.ModuleType("v3dref_ops")
    parent.__path__ = [os.path.join(REF, "ops")]
    sys.modules["v3dref_ops"] = parent
    try:
        pd = importlib.import_module("v3dref_ops.pairwise_distance")
        knn_mod = types.ModuleType("v3dref_ops.knn")

        def keops_knn(q_points, s_points, k):
            d = pd.pairwise_distance(q_points, s_points, squared=False, strict=True)
            vals, idx = d.topk(k, dim=-1, largest=False)
            return vals, idx

        def knn(q_points, s_points, k, return_distance=False, **_kw):
            vals, idx = keops_knn(q_points, s_points, k)
            return (vals, idx) if return_distance else idx

        knn_mod.keops_knn, knn_mod.knn = keops_knn, knn
        sys.modules["v3dref_ops.knn"] = knn_mod
        parent.knn = knn_mod
        ki = importlib.import_module("v3dref_ops.knn_interpolate")
        pcp = importlib.import_module("v3dref_ops.point_cloud_partition")
        rng = np.random.default_rng(20261121)
        out = {}
        # ---- knn_interpolate_pack_mode (:43-76) on a neighbour table with absent entries (index == M)
        n_q, n_s, kk, c = 260, 190, 6, 16
        q = rng.random((n_q, 3), dtype=np.float32)
        sp = rng.random((n_s, 3), dtype=np.float32)
        sf = rng.standard_normal((n_s, c)).astype(np.float32)
        d = np.linalg.norm(q[:, None].astype(np.float64) - sp[None].astype(np.float64), axis=-1)
        table = np.argsort(d, axis=1)[:, :kk].astype(np.int64)
        table[np.take_along_axis(d, table, 1) > 0.2] = n_s  # ordered neighbours, far ones absent (radius-search style)
        table[7] = n_s
        out.update(ki_q=q, ki_s=sp, ki_feats=sf, ki_table=table)
        for k_used in (None, 3):
            got = ki.knn_interpolate_pack_mode(torch.from_numpy(q), torch.from_numpy(sp), torch.from_numpy(sf),
                                               torch.from_numpy(table), k=k_used)
            out[f"ki_pack_k{k_used or 0}"] = got.numpy()
        # ---- knn_interpolate (:10-40), batch-free form, with and without a distance limit
        for lim in (None, 0.15):
            got = ki.knn_interpolate(torch.from_numpy(q), torch.from_numpy(sp), torch.from_numpy(sf), k=3,
                                     distance_limit=lim)
            out[f"ki_knn_lim{0 if lim is None else 1}"] = got.numpy()
        # ---- point_to_node_partition (:40-104)
        pts = (rng.random((700, 3)).astype(np.float32) * np.float32([3.0, 3.0, 2.5]))
        nodes = pts[rng.permutation(700)[:23]] + rng.normal(0, 0.05, (23, 3)).astype(np.float32)
        nodes = np.concatenate([nodes, np.float32([[50.0, 50.0, 50.0]])]).astype(np.float32)  # a node that owns no point
        out.update(p2n_points=pts, p2n_nodes=nodes)
        for tag, limit in (("all", None), ("lim", 20)):
            res = pcp.point_to_node_partition(torch.from_numpy(pts), torch.from_numpy(nodes), point_limit=limit,
                                              return_count=True)
            for key, t in zip(("p2n", "sizes", "masks", "knn_idx", "knn_masks"), res):
                out[f"p2n_{tag}_{key}"] = t.numpy()
        # ---- ball_query_partition (:107-124) on the stand-in knn
        res = pcp.ball_query_partition(torch.from_numpy(pts), torch.from_numpy(nodes), 0.45, 24, return_count=True)
        for key, t in zip(("knn_idx", "knn_masks", "masks", "sizes"), res):
            out[f"bqp_{key}"] = t.numpy()
        np.savez_compressed(os.path.join(HERE, "ref_knn_ops.npz"), **out)
    finally:
        torch.Tensor.cuda = orig_cuda
        for name in [k_ for k_ in sys.modules if k_.startswith("v3dref_ops")]:
            del sys.modules[name]
        _restore_modules(saved)


def make_pack_glue_fixtures():
    """The pure-torch pack-mode glue of the KPConv pyramid (models/graph_pyramid/graph_pyramid.py:9-63), run on the CPU:
      * ops/grid_subsample.py:13-46 `naive_grid_subsample_pack_mode` exactly as it is (no `.cuda()` in it) and :49-99
        `grid_subsample_pack_mode` with `Tensor.cuda` the identity (its one `.cuda()` is at :87);
      * ops/conversion.py:7-83 (`pack_to_batch`, `batch_to_pack`, `batch_to_pack_with_lengths`), same identity;
      * ops/radius_search.py:44-71 `radius_search_pack_mode` and :8-41 `naive_radius_search_pack_mode`.  The file imports
        pykeops at :2 and `keops_knn` from `.knn` at :4; pykeops is absent, so `.knn` is the stand-in of
        make_knn_ops_fixtures (the reference's own pairwise_distance + torch.topk) and `pykeops.torch.LazyTensor` is a
        name that raises when called (the two functions pinned here never touch it).  This pins the GLUE -- padding with
        1e10, per-cloud index offsets, the `> radius` mask -> sentinel, batch -> pack -- NOT pykeops' arithmetic, which stays
        "parity unpinned" (DESIGN.md 2).  Inputs keep every pair at least 1e-5 away from the radius and are tie-free.
    Empty clouds: both grid-subsample versions call `amin(0)` on the cloud's (0, 3) voxel tensor, which raises in torch, so
    the reference defines no result for them; the fixture records that (`gs_empty_cloud_raises`)."""
    import importlib
    import types

    orig_cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    saved = _synthetic_vision3d()
    saved.update({n: sys.modules.get(n) for n in ("pykeops", "pykeops.torch")})
    parent = types.ModuleType("v3dref_ops")
    parent.__path__ = [os.path.join(REF, "ops")]
    sys.modules["v3dref_ops"] = parent
    try:
        gs = importlib.import_module("v3dref_ops.grid_subsample")
        conv = importlib.import_module("v3dref_ops.conversion")
        rng = np.random.default_rng(20261201)
        out = {}
        # ---- grid subsampling: three clouds, negative coordinates, two voxel sizes
        lengths = np.asarray([900, 1300, 700], dtype=np.int64)
        pts = (rng.random((int(lengths.sum()), 3)).astype(np.float32) * np.float32([3.0, 3.0, 2.5])
               - np.float32([1.0, 0.5, 0.0]))
        out.update(gs_points=pts, gs_lengths=lengths)
        tp, tl = torch.from_numpy(pts), torch.from_numpy(lengths)
        for tag, vs in (("a", 0.3), ("b", 0.11)):
            sp, sl = gs.grid_subsample_pack_mode(tp.clone(), tl, vs)
            np_, nl = gs.naive_grid_subsample_pack_mode(tp.clone(), tl, vs)
            out[f"gs_{tag}_voxel"] = np.float64(vs)
            out[f"gs_{tag}_fast_points"], out[f"gs_{tag}_fast_lengths"] = sp.numpy(), sl.numpy()
            out[f"gs_{tag}_naive_points"], out[f"gs_{tag}_naive_lengths"] = np_.numpy(), nl.numpy()
        raised = []
        for fn in (gs.grid_subsample_pack_mode, gs.naive_grid_subsample_pack_mode):
            try:
                fn(tp[:1600].clone(), torch.tensor([900, 0, 700]), 0.3)
                raised.append(False)
            except (RuntimeError, IndexError):
                raised.append(True)
        out["gs_empty_cloud_raises"] = np.asarray(raised)
        # ---- conversion round trips
        c_lengths = np.asarray([5, 0, 9, 3], dtype=np.int64)
        packed = rng.standard_normal((int(c_lengths.sum()), 4)).astype(np.float32)
        out.update(cv_lengths=c_lengths, cv_packed=packed)
        for tag, max_len, fill in (("auto", None, 0.0), ("wide", 12, 1e10)):
            bt, bm = conv.pack_to_batch(torch.from_numpy(packed), torch.from_numpy(c_lengths), max_length=max_len,
                                        fill_value=fill)
            out[f"cv_{tag}_batch"], out[f"cv_{tag}_masks"] = bt.numpy(), bm.numpy()
            pk, pl = conv.batch_to_pack(bt, bm)
            out[f"cv_{tag}_repacked"], out[f"cv_{tag}_relengths"] = pk.numpy(), pl.numpy()
            pk2, pl2 = conv.batch_to_pack_with_lengths(bt, torch.from_numpy(c_lengths))
            out[f"cv_{tag}_repacked_with_lengths"] = pk2.numpy()
        pk3, pl3 = conv.batch_to_pack(bt)
        out["cv_nomask_lengths"] = pl3.numpy()
        out["cv_indices_from_lengths"] = conv._get_indices_from_lengths(torch.from_numpy(c_lengths), 12).numpy()
        # ---- radius search glue around the stand-in kNN
        pd = importlib.import_module("v3dref_ops.pairwise_distance")
        knn_mod = types.ModuleType("v3dref_ops.knn")

        def keops_knn(q_points, s_points, k):
            d = pd.pairwise_distance(q_points, s_points, squared=False, strict=True)
            return d.topk(k, dim=-1, largest=False)

        knn_mod.keops_knn = keops_knn
        sys.modules["v3dref_ops.knn"] = knn_mod
        parent.knn = knn_mod
        keops_pkg, keops_torch = types.ModuleType("pykeops"), types.ModuleType("pykeops.torch")
        keops_pkg.__path__ = []

        def _absent(*_a, **_k):
            raise RuntimeError("pykeops is not installed in this container")

        keops_torch.LazyTensor = _absent
        keops_pkg.torch = keops_torch
        sys.modules["pykeops"], sys.modules["pykeops.torch"] = keops_pkg, keops_torch
        rs = importlib.import_module("v3dref_ops.radius_search")
        ql = np.asarray([140, 60, 100], dtype=np.int64)
        sl_ = np.asarray([300, 45, 220], dtype=np.int64)
        radius, limit = 0.22, 24
        seed = 20261202
        while True:  # keep every pair clear of the radius so that fp32 rounding of any formulation cannot flip the mask
            r2 = np.random.default_rng(seed)
            q = r2.random((int(ql.sum()), 3), dtype=np.float32)
            s = r2.random((int(sl_.sum()), 3), dtype=np.float32)
            d = np.linalg.norm(q[:, None].astype(np.float64) - s[None].astype(np.float64), axis=-1)
            if np.abs(d - radius).min() > 1e-5:
                break
            seed += 1
        tq, ts = torch.from_numpy(q), torch.from_numpy(s)
        fast = rs.radius_search_pack_mode(tq, ts, torch.from_numpy(ql), torch.from_numpy(sl_), radius, limit)
        naive = rs.naive_radius_search_pack_mode(tq, ts, torch.from_numpy(ql), torch.from_numpy(sl_), radius, limit)
        assert torch.equal(fast, naive)
        out.update(rs_q=q, rs_s=s, rs_q_lengths=ql, rs_s_lengths=sl_, rs_radius=np.float64(radius),
                   rs_limit=np.int64(limit), rs_table=fast.numpy(), rs_seed=np.int64(seed))
        # ---- ops/knn.py:110-145 `knn_pack_mode` (and :30-107 `knn` on 3-D points) by the reference's own file: its
        # module-level `from pykeops.torch import LazyTensor` gets the raising name above, its `keops_knn` (:10-27, the
        # only user of LazyTensor) is replaced by the same stand-in.  Pins the glue -- padding to the longest cloud with
        # 1e10, indices inside the cloud, ascending distances, un-padding -- on tie-free clouds, not pykeops.
        del sys.modules["v3dref_ops.knn"]
        real_knn = importlib.import_module("v3dref_ops.knn")
        real_knn.keops_knn = keops_knn
        kq, ks = torch.from_numpy(q), torch.from_numpy(s)
        for k in (1, 3, 16):
            kd, ki = real_knn.knn_pack_mode(kq, ks, torch.from_numpy(ql), torch.from_numpy(sl_), k, return_distance=True)
            out[f"kp_k{k}_dist"], out[f"kp_k{k}_idx"] = kd.numpy(), ki.numpy()
        bq, bs = kq[:120].reshape(2, 60, 3), ks[:400].reshape(2, 200, 3)
        bd, bi = real_knn.knn(bq, bs, 5, return_distance=True)
        bd2, bi2 = real_knn.knn(bq, bs, 4, dilation=2, remove_nearest=True, return_distance=True)
        out.update(kb_dist=bd.numpy(), kb_idx=bi.numpy(), kb_dil_dist=bd2.numpy(), kb_dil_idx=bi2.numpy())
        np.savez_compressed(os.path.join(HERE, "ref_pack_glue.npz"), **out)
    finally:
        torch.Tensor.cuda = orig_cuda
        for name in [k_ for k_ in sys.modules if k_.startswith("v3dref_ops")]:
            del sys.modules[name]
        _restore_modules(saved)


def make_oracle_fixtures():
    rng = np.random.default_rng(7)
    # lattice: 6x6x6 integer grid scaled by 0.25 -> massive distance ties, exactly representable
    g = np.stack(np.meshgrid(np.arange(6), np.arange(6), np.arange(6), indexing="ij"), -1).reshape(-1, 3)
    lattice = (g.astype(np.float32) * 0.25)[None]  # (1, 216, 3)
    perm = rng.permutation(216)
    lattice_shuffled = lattice[:, perm]
    dup = np.repeat(unit_sphere_cloud(rng, 1, 40), 3, axis=1)  # every point three times
    rand100 = rng.random((2, 100, 3), dtype=np.float32)  # N=100 -> reference block 64 (not a divisor)
    clouds = {"lattice": lattice, "lattice_shuffled": lattice_shuffled, "dup": dup, "rand100": rand100}
    out = {}
    for name, pts in clouds.items():
        n = pts.shape[1]
        m = min(n, 48)
        out[f"{name}_pts"] = pts
        out[f"{name}_fps"] = oracle.furthest_point_sample(pts, m)
        qs = pts[:, : min(n, 32)]
        out[f"{name}_ball"] = oracle.ball_query(qs, pts, 8, 0.3)
        kd, ki = oracle.knn_points(qs, pts, 12)
        out[f"{name}_knn_d"], out[f"{name}_knn_i"] = kd, ki
        td, ti = oracle.three_nn(qs, pts)
        out[f"{name}_tnn_d"], out[f"{name}_tnn_i"] = td, ti
    # N < k and empty balls
    tiny = rng.random((1, 5, 3), dtype=np.float32)
    out["tiny_pts"] = tiny
    kd, ki = oracle.knn_points(tiny, tiny, 9)
    out["tiny_knn_d"], out["tiny_knn_i"] = kd, ki
    out["tiny_ball_empty"] = oracle.ball_query(tiny + 10.0, tiny, 4, 0.1)
    np.savez_compressed(os.path.join(HERE, "oracle_ties.npz"), **out)


if __name__ == "__main__":
    oracle.build()
    which = sys.argv[1:] or ["reference", "select", "matching", "voxelize", "pack", "oracle", "fps", "knn_ops", "pack_glue"]
    if "pack_glue" in which:
        make_pack_glue_fixtures()
    if "fps" in which:
        make_fps_fixtures()
    if "knn_ops" in which:
        make_knn_ops_fixtures()
    if "voxelize" in which:
        make_voxelize_fixtures()
    if "matching" in which:
        make_matching_fixtures()
    if "pack" in which:
        make_pack_ops_fixtures()
    if "reference" in which:
        make_reference_fixtures()
    if "select" in which:
        make_select_fixtures()
    if "oracle" in which:
        make_oracle_fixtures()
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)), "bytes")
