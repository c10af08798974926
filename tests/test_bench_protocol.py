"""bench.py's probe / agree / fall-back protocol for the path's only collective (mvs_gaussian_splatting_amd/dist.py:
negotiate_collectives) on a FAKE process group: N threads stand for N ranks, the agreement channel is a barrier-based
MIN all-reduce with a timeout -- a rank that skips the agreement (or enters it twice) makes the test fail instead of
hanging the first real 8-GPU run.  No GPU, no torch.distributed."""
import threading

import pytest

from mvs_gaussian_splatting_amd.dist import negotiate_collectives


class FakeGroup:
    """MIN all-reduce over `world` threads; every call must be matched by every rank (else BrokenBarrierError)."""

    def __init__(self, world):
        self.world = world
        self.barrier = threading.Barrier(world, timeout=20)
        self.values = [None] * world
        self.calls = [0] * world

    def agree_min(self, rank, flag):
        self.calls[rank] += 1
        self.values[rank] = flag
        self.barrier.wait()
        out = min(self.values)
        self.barrier.wait()          # nobody overwrites `values` before everybody has read them
        return out


def _run(world, probe_of_rank, backend="nccl"):
    grp = FakeGroup(world)
    plans, errors = [None] * world, []

    def rank_main(r):
        try:
            plans[r] = negotiate_collectives(world, backend, lambda: probe_of_rank(r), lambda f: grp.agree_min(r, f))
        except Exception as ex:  # noqa: BLE001
            errors.append((r, repr(ex)))

    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=30)
        assert not t.is_alive(), "a rank hung in the negotiation"
    assert not errors, errors
    return plans, grp


def _same(plans):
    key = lambda p: (p.cpu_collectives, p.rccl_ranks, p.rccl_failed, p.backend_note)  # noqa: E731
    assert len({key(p) for p in plans}) == 1, [key(p) for p in plans]
    return plans[0]


@pytest.mark.parametrize("world", [2, 4, 8])
def test_all_ranks_up_keeps_rccl(world):
    plans, grp = _run(world, lambda r: float(world))
    p = _same(plans)
    assert not p.cpu_collectives and p.rccl_ranks == world and not p.rccl_failed and p.backend_note == "nccl"
    assert grp.calls == [1] * world


@pytest.mark.parametrize("world,bad", [(2, {1}), (8, {3}), (8, {0, 7}), (4, {0, 1, 2, 3})])
def test_a_failed_probe_on_any_rank_switches_every_rank(world, bad):
    def probe(r):
        if r in bad:
            raise RuntimeError("NCCL error: duplicate GPU detected")
        return float(world)
    plans, grp = _run(world, probe)
    p = _same(plans)
    assert p.cpu_collectives and p.rccl_failed and p.rccl_ranks is None and p.backend_note.startswith("gloo")
    assert grp.calls == [1] * world                      # the healthy ranks took part in the agreement too
    for r in range(world):                               # ... and each rank can say why
        assert ("duplicate GPU" in plans[r].why) == (r in bad)


def test_a_short_communicator_is_a_failed_probe():
    """An all-reduce of ones that returns less than the world size means RCCL formed a smaller communicator."""
    plans, _ = _run(4, lambda r: 2.0 if r < 2 else 4.0)
    p = _same(plans)
    assert p.rccl_failed and p.cpu_collectives
    assert "returned 2.0, expected 4" in plans[0].why


def test_single_process_and_explicit_gloo_need_no_agreement():
    def boom(*a):
        raise AssertionError("must not be called")
    p = negotiate_collectives(1, "nccl", boom, boom)
    assert not p.cpu_collectives and not p.rccl_failed and p.rccl_ranks is None
    p = negotiate_collectives(4, "gloo", boom, boom)
    assert p.cpu_collectives and not p.rccl_failed and p.backend_note == "gloo"


def test_byte_models_of_the_roofline_objects():
    """bench.py prices the per-Gaussian kernels on the bytes a frame has to move (model v2) and keeps SURVEY §8(d)'s
    model v1 beside it: v2 never exceeds v1 by more than the instance rows, equals v1's shape when everything is visible,
    and charges a culled Gaussian 44 bytes in the forward."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    P, M, W, H = 6_000_000, 16, 1920, 1080
    v1 = bench.algorithmic_bytes(P, M, 8_192_108, W, H, 6)
    assert v1["preprocess_fwd"] == P * (44 + 12 * M + 75) and v1["render_fwd"] == 40 * 8_192_108 + 20 * W * H
    v2_all = bench.bytes_really_moved(P, M, P, 0, W, H)
    assert v2_all["preprocess_fwd"] == v1["preprocess_fwd"]                  # everything visible: the same bytes
    v2 = bench.bytes_really_moved(P, M, 3_880_905, 8_192_108, W, H)
    assert v2["preprocess_fwd"] == 44 * P + (12 * M + 75) * 3_880_905 < v1["preprocess_fwd"]
    assert bench.bytes_really_moved(P, M, 0, 0, W, H)["preprocess_fwd"] == 44 * P      # nothing visible: 44 B per Gaussian
    assert v2["preprocess_bwd"] < v1["preprocess_bwd"] + 49 * 8_192_108


def _bench_module():
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    return bench


@pytest.mark.parametrize("world", [1, 2, 4, 8])
def test_eight_views_are_partitioned_round_robin_and_summarised_from_the_slowest_rank(world):
    """rank r renders views {r, r + N, ...} of the eight; every view exactly once; the whole-job numbers of the
    eight-view loop come from the slowest rank, the per-rank fields keep every rank's own time per view."""
    bench = _bench_module()
    parts = [bench.c5_views_of_rank(r, world) for r in range(world)]
    assert sorted(v for p in parts for v in p) == list(range(8))
    assert all(p == list(range(r, 8, world)) for r, p in enumerate(parts))
    assert all(len(p) == 8 // world for p in parts)
    W, H, rounds = 1920, 1080, 5
    fwd = [0.010 * (1 + 0.1 * r) for r in range(world)]          # seconds over `rounds` rounds, rank r a little slower
    train = [0.025 * (1 + 0.2 * r) for r in range(world)]
    s = bench.c5_summary(fwd, train, rounds, world, W, H)
    assert s["views_per_rank"] == [8 // world] * world and s["views"] == 8 and s["rounds"] == rounds
    assert s["fwd_ms_per_round"] == round(max(fwd) / rounds * 1e3, 3)
    assert s["train_ms_per_round"] == round(max(train) / rounds * 1e3, 3)
    assert s["fwd_mpixels_per_s"] == round(8 * W * H * rounds / max(fwd) / 1e6, 2)
    assert len(s["per_rank_fwd_ms_per_view"]) == world
    assert s["per_rank_train_ms_per_view"][0] == round(train[0] / rounds / (8 // world) * 1e3, 3)


def test_issue_roof_model_prices_the_counted_classes_at_their_measured_rates():
    """frac_of_issue_roof = (2.4 full-rate + 8.2 transcendental + rest_cost rest) cycles / (1024 SIMDs x clock x t)."""
    bench = _bench_module()
    mix = {"SQ_INSTS_VALU": 1000, "SQ_INSTS_VALU_ADD_F32": 100, "SQ_INSTS_VALU_MUL_F32": 100, "SQ_INSTS_VALU_FMA_F32": 200,
           "SQ_INSTS_VALU_TRANS_F32": 50, "SQ_INSTS_VALU_INT32": 150, "SQ_INSTS_VALU_INT64": 0, "SQ_INSTS_VALU_CVT": 0}
    m = bench.issue_model(mix, 4.0, 1e-6, 2.0e9)
    cycles = 2.4 * 550 + 8.2 * 50 + 4.0 * 400
    assert m["full_rate"] == 550 and m["transcendental"] == 50 and m["rest"] == 400
    assert m["valu_issue_cycles"] == int(cycles)
    assert m["frac_of_issue_roof"] == round(cycles / (1024 * 2.0e9 * 1e-6), 4)
    assert bench.issue_model({}, 4.0, 1e-6, 2.0e9) is None and bench.issue_model(mix, 4.0, 0.0, 2.0e9) is None


# ---- `python bench.py --gpus N` as its own launcher ---------------------------------------------------
_FAKE_RANK = '''
import json, os, sys
import torch, torch.distributed as dist
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
assert os.environ["MASTER_ADDR"] == "127.0.0.1"
mode = sys.argv[sys.argv.index("--mode") + 1] if "--mode" in sys.argv else "ok"
dist.init_process_group(backend="gloo")
t = torch.tensor([1.0, float(rank)])
dist.all_reduce(t)
print("banner of a native library, rank", rank)           # noise on stdout: must not reach the launcher's stdout
if mode == "fail" and rank == world - 1:
    sys.exit(7)
if rank == 0 and mode != "silent":
    print(json.dumps({"metric": "fake", "n_gpus": world, "ranks_seen": t[0].item(), "rank_sum": t[1].item(),
                      "share": os.environ.get("GSR_BENCH_SHARE_GPU"), "argv": sys.argv[1:]}))
dist.barrier()
dist.destroy_process_group()
'''


def _launch(tmp_path, world, mode, gpus_visible):
    import io
    import json
    bench = _bench_module()
    script = tmp_path / "fake_rank.py"
    script.write_text(_FAKE_RANK)
    out = io.StringIO()
    rc = bench.launch_ranks(world, ["--gpus", str(world), "--mode", mode], script=str(script), out=out, timeout=240,
                            gpus_visible=gpus_visible)
    lines = [ln for ln in out.getvalue().splitlines() if ln.strip()]
    return rc, [json.loads(ln) for ln in lines]


@pytest.mark.parametrize("world", [2, 4])
def test_plain_command_launches_n_fresh_ranks_and_relays_one_line(tmp_path, world):
    """`python bench.py --gpus N` without WORLD_SIZE: N child processes under torch.distributed.run (gloo here), rank 0's
    single JSON line on the launcher's stdout and nothing else, exit code 0; the arguments reach every rank unchanged."""
    rc, lines = _launch(tmp_path, world, "ok", gpus_visible=8)
    assert rc == 0 and len(lines) == 1
    d = lines[0]
    assert d["n_gpus"] == world and d["ranks_seen"] == world and d["rank_sum"] == sum(range(world))
    assert d["argv"] == ["--gpus", str(world), "--mode", "ok"]
    assert d["share"] is None                       # enough devices: every rank gets its own


def test_launcher_marks_a_one_card_rehearsal_and_reports_failures(tmp_path):
    rc, lines = _launch(tmp_path, 2, "ok", gpus_visible=1)
    assert rc == 0 and lines[0]["share"] == "1"     # fewer devices than ranks: the ranks are told to share cuda:0
    rc, lines = _launch(tmp_path, 2, "fail", gpus_visible=8)
    assert rc != 0 and lines == []                  # a failed rank: non-zero exit, no result line handed on
    rc, lines = _launch(tmp_path, 2, "silent", gpus_visible=8)
    assert rc == 1 and lines == []                  # every rank exited 0 but nobody printed a result


def test_result_line_is_picked_among_noise():
    bench = _bench_module()
    txt = 'NCCL version 2.x\n{"not": "it"}\n{"metric": "m", "value": 1}\ntrailing banner\n'
    assert bench.pick_result_line(txt) == '{"metric": "m", "value": 1}'
    assert bench.pick_result_line("nothing here\n{broken json}\n") is None


def test_bench_refuses_a_world_size_that_contradicts_gpus(monkeypatch):
    """Under a torchrun environment `--gpus` must equal WORLD_SIZE (the driver passes both); a mismatch is an error, not a
    silent single-rank run."""
    import subprocess
    import sys
    import os
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--no-rccl-probe"], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert r.returncode != 0 and b"WORLD_SIZE=2 but --gpus 1" in r.stderr
