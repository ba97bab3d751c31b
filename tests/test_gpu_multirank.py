"""The N > 1 path on real kernels.  One process per rank, started as fresh child processes (never an exec of a process that
touched the GPU):
  * on a one-GPU box the ranks share cuda:0 and talk over gloo (device tensors staged through the host by
    sharding._all_gather_rows): the HIP search / merge / selection kernels run under the real collective plumbing;
  * with two or more GPUs visible, `bench.py --gpus 2` starts its own ranks over RCCL (skipped otherwise).
Reference counterpart: vision3d/utils/distributed.py:11-18 (NCCL process group, one rank per device)."""
import json
import os
import socket
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _json_line(text):
    lines = [ln for ln in text.splitlines() if ln.startswith("{")]
    assert lines, text[-2000:]
    return json.loads(lines[-1])


def _worker(rank, world, port):
    """Reference-set sharding with the default (HIP) kernels over gloo; every rank checks against its own single-shot run."""
    import torch
    import torch.distributed as dist

    sys.path.insert(0, REPO)
    from vision3d_amd import sharding

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dev = torch.device("cuda", 0)
        g = torch.Generator().manual_seed(7)  # same data on every rank
        q = torch.nn.functional.normalize(torch.randn((3000, 64), generator=g), dim=1)
        refs = torch.nn.functional.normalize(torch.randn((5001, 64), generator=g), dim=1)
        refs[:500] = q[:500] + 0.01 * torch.randn((500, 64), generator=g)  # planted matches
        refs[4000] = refs[10]  # an exact duplicate in another shard: the lower global index must come first
        q, refs = q.to(dev), refs.to(dev)
        n_ref = refs.shape[0]
        b, e = sharding.batch_bounds(n_ref, world, rank)
        local = refs[b:e].contiguous()
        for k in (1, 3):
            d, i = sharding.knn_reference_sharded(q, local, k, b)
            fd, fi = sharding._hip_search(q, refs, k, 0)
            assert torch.equal(i, fi) and torch.equal(d, fd), f"sharded kNN differs from single shot (k={k})"
            for mutual in (True, False):
                got = sharding.mutual_matching_sharded(q, local, k, b, n_ref, mutual=mutual)
                (rd, ri), (cd, ci) = (fd, fi), sharding._hip_search(refs, q, k, 0)
                want = sharding._hip_select(ri, rd, ci, cd, mutual)
                assert all(torch.equal(a, w) for a, w in zip(got, want)), f"sharded matching differs (k={k}, mutual={mutual})"
                if mutual and k == 1:
                    assert int((got[0] < 500).sum()) >= 450  # the planted pairs are found
        # ---- pack-mode batch sharding (BASELINE config 4): real radius-search / grid-subsample kernels on this rank's
        # clouds; the gathered, globalized tables equal the single-process table; so does every stage of the pyramid
        from vision3d_amd import ops

        g = torch.Generator().manual_seed(9)
        lens = torch.tensor([3000, 0, 1700, 2500, 900])
        pts = (torch.rand((int(lens.sum()), 3), generator=g) * torch.tensor([3.0, 3.0, 2.5])).to(dev)
        dlens = lens.to(dev)
        table, qs, ss = sharding.radius_search_pack_sharded(pts, pts, lens, lens, 0.2, 20)
        whole = sharding.gather_pack_tables(table, ss, pts.shape[0])
        assert torch.equal(whole, ops.radius_search_pack_mode(pts, pts, dlens, dlens, 0.2, 20)), "sharded radius search"
        mine = ops.build_grid_and_radius_graph_pyramid(qs.points.contiguous(), dlens[qs.cloud_begin:qs.cloud_end].contiguous(),
                                                       3, 0.1, 0.25, [20, 18, 16])
        full = ops.build_grid_and_radius_graph_pyramid(pts, dlens, 3, 0.1, 0.25, [20, 18, 16])
        for i in range(3):
            cur = sharding.shard_pack(full["points"][i], full["lengths"][i].cpu(), world, rank)
            assert torch.equal(mine["points"][i], cur.points) and torch.equal(mine["lengths"][i].cpu(), cur.lengths)
            n_cur, tot_cur = cur.row_end - cur.row_begin, full["points"][i].shape[0]
            assert torch.equal(sharding.globalize_pack_table(mine["neighbors"][i], n_cur, cur.row_begin, tot_cur),
                               full["neighbors"][i][cur.row_begin:cur.row_end]), f"pyramid neighbors, stage {i}"
            if i < 2:
                sub = sharding.shard_pack(full["points"][i + 1], full["lengths"][i + 1].cpu(), world, rank)
                n_sub, tot_sub = sub.row_end - sub.row_begin, full["points"][i + 1].shape[0]
                assert torch.equal(sharding.globalize_pack_table(mine["subsampling"][i], n_cur, cur.row_begin, tot_cur),
                                   full["subsampling"][i][sub.row_begin:sub.row_end]), f"pyramid subsampling, stage {i}"
                assert torch.equal(sharding.globalize_pack_table(mine["upsampling"][i], n_sub, sub.row_begin, tot_sub),
                                   full["upsampling"][i][cur.row_begin:cur.row_end]), f"pyramid upsampling, stage {i}"
        ones = torch.ones(1, device=dev)
        dist.all_reduce(ones)
        assert int(ones.item()) == world
        print(f"rank {rank} OK", flush=True)
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_reference_set_sharding_two_ranks_real_kernels():
    port = _free_port()
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "worker", str(r), "2", str(port)],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=110)[0] for p in procs]
    for r, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and f"rank {r} OK" in out, out[-3000:]


@pytest.mark.gpu
def test_bench_starts_its_own_ranks_c5_refset_on_one_gpu():
    """`bench.py --gpus 2` without a launcher: the parent starts two ranks; here they share the GPU and use gloo."""
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--share-gpu", "--backend", "gloo",
                          "--workload", "c5_refset", "--steps", "1", "--warmup", "0"], capture_output=True, text=True,
                         timeout=110)
    assert out.returncode == 0, out.stderr[-3000:]
    line = _json_line(out.stdout)
    assert line["n_gpus"] == 2 and line["rccl_world_size"] == 2 and line["equals_single_gpu_result"] is True
    assert line["scaling"] == "strong" and line["allgather_bytes_per_rank"] == 100000 * 3 * 12


@pytest.mark.gpu
def test_bench_starts_its_own_ranks_c4_pyramid_on_one_gpu():
    """BASELINE config 4 under pack-mode batch sharding, two ranks sharing the GPU over gloo: a line is printed under N > 1
    and the per-rank tables equal the single-process ones."""
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--share-gpu", "--backend", "gloo",
                          "--workload", "c4_pyramid", "--steps", "2", "--warmup", "1"], capture_output=True, text=True,
                         timeout=110)
    assert out.returncode == 0, out.stderr[-3000:]
    line = _json_line(out.stdout)
    assert line["n_gpus"] == 2 and line["equals_single_gpu_result"] is True and line["scaling"] == "weak"
    assert line["radius_queries_per_step_all_ranks"] > 2 * 16 * 30000 and line["value"] > 0


@pytest.mark.gpu
def test_bench_gpus2_over_rccl():
    import torch

    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (the driver's multi-GPU node); one-GPU boxes run the gloo rehearsal above")
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2",
                          "--no-cpu-baseline"], capture_output=True, text=True, timeout=115)
    assert out.returncode == 0, out.stderr[-3000:]
    line = _json_line(out.stdout)
    assert line["n_gpus"] == 2 and line["rccl_world_size"] == 2
    assert line["per_config"]["c5_matching_refset_sharded"]["equals_single_gpu_result"] is True
    assert line["per_config"]["c4_kpconv_pyramid_sharded"]["equals_single_gpu_result"] is True


def test_bench_refuses_more_ranks_than_gpus():
    """Parent mode, no GPU touched: asking for more ranks than visible devices is an error, not a silent single rank."""
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "4096"], capture_output=True, text=True,
                         timeout=110)
    assert out.returncode != 0 and "--gpus 4096" in (out.stderr + out.stdout)


if __name__ == "__main__" and len(sys.argv) == 5 and sys.argv[1] == "worker":
    _worker(int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]))
