"""Differential fuzzing of the cell-list query kernels against the CPU oracle (GPU box; test infrastructure, not shipped):
random cloud shapes, sizes, distributions and k for knn_points (per-workgroup cell list), knn_lex, knn_pack (pair by pair
and global cell list) and radius_search on a caller-held grid.  usage: python tools/fuzz_queries.py [seconds] [seed]"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import oracle  # noqa: E402
from vision3d_amd import ext  # noqa: E402

DEV = "cuda:0"
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)  # noqa: E731
N_ = lambda t: t.cpu().numpy()  # noqa: E731


def cloud(rng, n):
    kind = rng.integers(0, 7)
    if kind == 0:
        p = rng.random((n, 3))
    elif kind == 1:  # clusters
        c = rng.random((rng.integers(1, 8), 3)) * rng.choice([1.0, 10.0, 100.0])
        p = c[rng.integers(0, len(c), n)] + rng.normal(0, rng.choice([1e-3, 0.05, 0.5]), (n, 3))
    elif kind == 2:  # lattice with ties
        side = int(np.ceil(n ** (1 / 3))) + 1
        g = np.stack(np.meshgrid(*[np.arange(side)] * 3, indexing="ij"), -1).reshape(-1, 3)
        p = g[rng.permutation(len(g))[:n]] * 0.25
    elif kind == 3:  # flat / line
        p = rng.random((n, 3))
        p[:, rng.integers(0, 3)] = 0.5
        if rng.random() < 0.3:
            p[:, rng.integers(0, 3)] = -1.0
    elif kind == 4:  # duplicates
        base = rng.random((max(n // 4, 1), 3))
        p = base[rng.integers(0, len(base), n)]
    elif kind == 5:  # surface of a sphere
        v = rng.standard_normal((n, 3))
        p = v / np.linalg.norm(v, axis=1, keepdims=True) * rng.choice([0.5, 5.0])
    else:  # anisotropic box far from the origin
        p = rng.random((n, 3)) * np.array([50.0, 0.5, 3.0]) + np.array([1000.0, -20.0, 3.0])
    return p.astype(np.float32)


def queries(rng, s, m):
    kind = rng.integers(0, 3)
    if kind == 0:
        q = s[rng.integers(0, len(s), m)]
    elif kind == 1:
        q = s[rng.integers(0, len(s), m)] + rng.normal(0, 0.01, (m, 3)).astype(np.float32)
    else:
        lo, hi = s.min(0), s.max(0)
        q = lo + (rng.random((m, 3)) * 1.6 - 0.3) * np.maximum(hi - lo, 1e-3)
    return q.astype(np.float32)


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    oracle.set_threads(oracle.max_threads())
    t_end = time.time() + budget
    n_cases, bad = 0, 0
    routes = [0, 0]  # knn_pack calls searched pair by pair / on the cell list
    from vision3d_amd import _lib
    lib = _lib.lib()
    while time.time() < t_end:
        which = rng.integers(0, 4)
        tag = None
        if which == 0:  # knn_points / knn_lex, batch mode, sizes around the LDS cell list's range
            B = int(rng.integers(1, 4))
            N = int(rng.choice([rng.integers(1, 300), rng.integers(1800, 2300), rng.integers(2048, 4097), rng.integers(4000, 4200)]))
            M = int(rng.choice([rng.integers(1, 70), rng.integers(60, 400)]))
            k = int(rng.choice([1, 2, 5, 16, 31, 32, 63, 64, 100]))
            s = np.stack([cloud(rng, N) for _ in range(B)])
            q = np.stack([queries(rng, s[b], M) for b in range(B)])
            tag = f"knn_points B{B} N{N} M{M} k{k}"
            wd, wi = oracle.knn_points(q, s, k)
            d = torch.full((B, M, k), -1.0, device=DEV)
            i = torch.full((B, M, k), -1, dtype=torch.long, device=DEV)
            ext.knn_points(T(q), T(s), d, i, k)
            ok = np.array_equal(N_(i), wi) and np.array_equal(N_(d), wd)
            ld, li = oracle.knn_pack(q.reshape(-1, 3), s.reshape(-1, 3), np.full(B, M), np.full(B, N), k)
            ext.knn_lex(T(q), T(s), d, i, k)
            ok = ok and np.array_equal(N_(i).reshape(-1, k), li) and np.array_equal(N_(d).reshape(-1, k), ld)
        elif which in (1, 2):  # knn_pack: small clouds (pair by pair) or big ones (cell list)
            B = int(rng.integers(1, 5))
            big = which == 2
            sl = np.array([int(rng.integers(2048, 7000)) if big else int(rng.integers(0, 900)) for _ in range(B)])
            if big and rng.random() < 0.3:
                sl[rng.integers(0, B)] = int(rng.integers(0, 40))  # a tiny cloud among big ones
                sl[rng.integers(0, B)] += 2048 * B                  # ... keep the average up
            ql = np.array([int(rng.integers(0, 300)) for _ in range(B)])
            k = int(rng.choice([1, 2, 3, 4, 8, 16, 38, 64] + ([] if big else [100])))
            sc = [cloud(rng, int(n)) if n else np.zeros((0, 3), np.float32) for n in sl]
            qc = [queries(rng, sc[b], int(ql[b])) if sl[b] else rng.random((int(ql[b]), 3), dtype=np.float32)
                  for b in range(B)]
            s, q = np.concatenate(sc), np.concatenate(qc)
            if len(q) == 0:
                continue
            tag = f"knn_pack {'cell-list' if big else 'pairwise'} sl{sl.tolist()} ql{ql.tolist()} k{k}"
            wd, wi = oracle.knn_pack(q, s, ql, sl, k)
            # the cell list is only taken with enough queries to pay for its build: repeat every cloud's queries
            rep = -(-250000 // len(q)) if big else 1
            q_rep = np.concatenate([np.tile(c, (rep, 1)) for c in qc])
            route = lib.v3d_knn_pack_uses_cell_list(B, len(q_rep), len(s), k)
            routes[route] += 1
            d = torch.full((len(q_rep), k), -1.0, device=DEV)
            i = torch.full((len(q_rep), k), -1, dtype=torch.long, device=DEV)
            ext.knn_pack(T(q_rep), T(s), T(ql * rep), T(sl), d, i, k)
            gd, gi = N_(d), N_(i)
            ok, qo, ro = True, 0, 0
            for n in ql:
                n = int(n)
                for r in range(rep):
                    rows = slice(ro + r * n, ro + (r + 1) * n)
                    ok = ok and np.array_equal(gi[rows], wi[qo:qo + n]) and np.array_equal(gd[rows], wd[qo:qo + n])
                qo += n
                ro += n * rep
        else:  # radius search: whole call and caller-held grid at a smaller radius
            B = int(rng.integers(1, 4))
            sl = np.array([int(rng.integers(0, 3000)) for _ in range(B)])
            ql = np.array([int(rng.integers(0, 300)) for _ in range(B)])
            sc = [cloud(rng, int(n)) if n else np.zeros((0, 3), np.float32) for n in sl]
            qc = [queries(rng, sc[b], int(ql[b])) if sl[b] else rng.random((int(ql[b]), 3), dtype=np.float32)
                  for b in range(B)]
            s, q = np.concatenate(sc), np.concatenate(qc)
            if len(q) == 0 or len(s) == 0:
                continue
            ext_ = float(np.maximum(s.max(0) - s.min(0), 1e-3).max())
            cell_r = float(ext_ * rng.choice([0.02, 0.1, 0.5]))
            radius = float(cell_r * rng.choice([1.0, 0.5, 0.25]))
            limit = int(rng.choice([1, 8, 38, 64, 100]))
            tag = f"radius sl{sl.tolist()} ql{ql.tolist()} cell_r{cell_r:.4g} r{radius:.4g} limit{limit}"
            want = oracle.radius_search_pack(q, s, ql, sl, radius, limit)
            out = torch.full((len(q), limit), -1, dtype=torch.long, device=DEV)
            ext.radius_search_pack(T(q), T(s), T(ql), T(sl), out, radius, limit)
            ok = np.array_equal(N_(out), want)
            grid = ext.radius_grid_build(T(s), T(sl), cell_r)
            out.fill_(-1)
            ext.radius_search_grid(T(q), T(ql), grid, out, radius, limit)
            ok = ok and np.array_equal(N_(out), want)
        n_cases += 1
        if not ok:
            bad += 1
            print("MISMATCH", tag, flush=True)
        if n_cases % 25 == 0:
            print(f"{n_cases} cases, {bad} mismatches", flush=True)
    print(f"done: {n_cases} cases, {bad} mismatches (seed {seed}); knn_pack routes: {routes[0]} pair by pair, "
          f"{routes[1]} on the cell list", flush=True)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
