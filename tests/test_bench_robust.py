"""bench.py --gpus N must print exactly ONE parsable JSON line whatever the process group does
(the 8-GPU scaling run is the driver's, never the builder's: a traceback there wastes it).

The script's control flow runs here on CPU (PPK_BENCH_FAKE=1: gloo, a stand-in for the kernels, the
line says so in `data`), two ranks under torch.distributed.run, with a collective made to fail
(PPK_BENCH_INJECT=isend:<after n calls>) or a rank made to stop answering (hang:<n>).
"""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(inject, extra=()):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, PPK_BENCH_FAKE="1", MASTER_ADDR="127.0.0.1")
    env.pop("PPK_BENCH_INJECT", None)
    if inject:
        env["PPK_BENCH_INJECT"] = inject
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
           "--gpus", "2", "--steps", "3", "--warmup", "1", "--genomes", "1500", "--no-cpu",
           "--collective-timeout", "8", "--watchdog-s", "25"] + list(extra)
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=180, cwd=ROOT)
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, (p.stdout.decode()[-2000:], p.stderr.decode()[-2000:])
    return json.loads(lines[0])


def _check_common(d):
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["unit"] == "pairs/s"
    assert "FAKE" in d["data"]
    mg = d["multi_gpu"]
    per = mg["per_rank_compute_only_ms_per_step"]
    assert len(per) == 2 and all(x is not None and x > 0 for x in per)
    assert sum(mg["per_rank_band_pairs"]) == d["config"]["pairs"] == 1500 * 1499 // 2
    return mg


def test_clean_run_prints_the_gathered_value():
    d = _run(None)
    mg = _check_common(d)
    assert "error" not in mg and "value_note" not in d
    assert sorted(mg["chunks_probe_ms_per_step"]) == ["1", "2", "4", "8"]      # the set-up probed the sub-band counts
    assert d["value"] > 0 and d["ms_per_step"] > 0
    assert abs(sum(mg["band_shares"]) - 1.0) < 1e-3
    # the scaling run describes itself: the backend really connected both ranks, both transports are on the line
    # (the FAKE run has no IPC window: the peer-store entry says why), `value` names its transport
    assert mg["rccl"]["backend"] == "gloo" and mg["rccl"]["world_size"] == 2 and mg["rccl"]["allreduce_of_ones"] == 2.0
    assert mg["gathered"]["value"] == d["value"] and mg["gathered"]["ms_per_step"] == d["ms_per_step"]
    assert isinstance(mg["peer_store"], dict) and mg["peer_store"].get("available") is False
    assert d["value_transport"] == "gather" and mg["transport_requested"] == "best"


@pytest.mark.parametrize("inject", ["isend:0", "isend:7", "isend:30", "isend:60"])
def test_failing_send_recv_still_yields_one_line_with_per_rank_compute(inject):
    """The first gathered step, the rebalance probe, the sub-band probe or the timed loop loses batch_isend_irecv: the line
    carries multi_gpu.error, every rank's compute-only ms per step (measured before any collective,
    handed over through files) and a value that says it excludes the gather."""
    d = _run(inject)
    mg = _check_common(d)
    assert "injected failure" in mg["error"]
    assert d["value"] > 0 and "EXCLUDES the gather" in d["value_note"]
    assert d["value_transport"].startswith("compute_only") and mg["gathered"]["available"] is False
    assert mg["rccl"]["allreduce_of_ones"] == 2.0          # the group itself had formed
    assert d["ms_per_step"] == max(mg["per_rank_compute_only_ms_per_step"])


def test_a_rank_that_stops_answering_ends_in_a_line_not_a_hang():
    d = _run("hang:5")
    mg = _check_common(d)
    assert mg["error"]
    assert d["value"] > 0


def test_even_bands_flag_skips_the_rebalance():
    d = _run(None, ["--even-bands"])
    mg = _check_common(d)
    assert mg["bands"].startswith("equal (--even-bands)") and "error" not in mg


def test_no_collective_legs_survive_a_process_group_that_never_forms():
    """Round-3 verdict: `multi_gpu.host_call` and `config5.host_call` -- one process driving all N GPUs, no
    collective -- are measured by rank 0 BEFORE torch.distributed is initialised, so a failing init_process_group
    still leaves both on the line (FAKE mode: stand-ins, the control flow is what is tested)."""
    d = _run("init")
    mg = _check_common(d)
    assert "injected failure of init_process_group" in mg["error"]
    assert mg["host_call"] == {"fake": True, "devices": [0, 1]}
    assert d["config5"]["host_call"] == {"fake": True}
    assert d["value"] > 0 and "EXCLUDES the gather" in d["value_note"]


def test_clean_run_carries_the_solo_legs_too():
    d = _run(None)
    assert d["multi_gpu"]["host_call"] == {"fake": True, "devices": [0, 1]}


def test_recorded_traffic_is_refused_when_the_library_was_built_from_other_sources(tmp_path):
    """profiles/pmc_traffic.json is a RECORDED figure (rocprofv3 --pmc passes); bench.py may print it only beside a
    kernel built from the sources it was measured on (the stamp = ppk_version()'s source hash)."""
    sys.path.insert(0, ROOT)
    import bench
    from poppunk_amd import _lib
    built = _lib.source_hash()
    assert built and built == _lib.sources_hash_now(), "libppk_hip.so is older than its sources: rebuild"
    good = tmp_path / "t.json"
    good.write_text(json.dumps({"n10000": 123.0, "source": "x", "src_hash": built}))
    assert bench.recorded_traffic(10000, str(good)) == (123.0, "x", False)
    stale = tmp_path / "s.json"
    stale.write_text(json.dumps({"n10000": 123.0, "source": "x", "src_hash": "0123456789abcdef"}))
    t, src, is_stale = bench.recorded_traffic(10000, str(stale))
    assert t is None and is_stale and "0123456789abcdef" in src
    unstamped = tmp_path / "u.json"
    unstamped.write_text(json.dumps({"n10000": 123.0, "source": "x"}))
    assert bench.recorded_traffic(10000, str(unstamped))[0] is None and bench.recorded_traffic(10000, str(unstamped))[2]
    assert bench.recorded_traffic(10000, str(tmp_path / "absent.json")) == (None, None, False)


def test_a_rank_that_dies_still_leaves_rank_0s_line():
    """torch.distributed.run answers a rank's non-zero exit with SIGTERM to the others; rank 0 may be inside a
    collective at that moment.  The signal reaches a thread of its own through a wake-up pipe and the line goes
    out with the error and rank 0's own compute-only figure."""
    d = _run("die")
    assert d["n_gpus"] == 2 and "FAKE" in d["data"]
    mg = d["multi_gpu"]
    assert mg["error"]          # "SIGTERM in phase ..." and / or the broken collective rank 0 was in
    assert mg["per_rank_compute_only_ms_per_step"][0] is None or mg["per_rank_compute_only_ms_per_step"][0] > 0
    assert mg["host_call"] == {"fake": True, "devices": [0, 1]}          # measured before the process group


def test_the_line_takes_the_faster_of_the_two_transports_only_when_the_matrices_were_identical(monkeypatch):
    """N > 1: the gathered steps and the peer-store steps (every rank's kernel stores its band into rank 0's matrix)
    are both timed over K steps; `value` is the peer-store figure only if that leg ran, produced the gathered matrix
    bit for bit and was faster -- the other figure stays on the line either way."""
    sys.path.insert(0, ROOT)
    import bench
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2", "--steps", "10", "--warmup", "1", "--genomes", "2000"])
    args = bench.parse()

    def line(ps):
        rep = bench.Report(0, 2, args)
        rep.fields.update({"n": 2000, "elapsed": 0.05, "compute_ms_max": 2.0, "peer_store": ps})
        return bench.build_line(rep)

    pairs = 2000 * 1999 // 2
    good = {"available": True, "identical_to_gathered": True, "ms_per_step": 2.5, "pairs_per_s": pairs / 2.5e-3}
    d = line(good)
    assert d["ms_per_step"] == 2.5 and d["value"] == good["pairs_per_s"]
    assert d["multi_gpu"]["gathered"]["ms_per_step"] == pytest.approx(5.0)
    assert "peer-store" in d["multi_gpu"]["transport"] and "stores its band" in d["config"]["parallelism"]
    assert d["value_transport"] == "peer_store"
    for ps in (dict(good, identical_to_gathered=False), dict(good, ms_per_step=7.0), {"available": False, "why": "x"}, None):
        d = line(ps)
        assert d["ms_per_step"] == pytest.approx(5.0) and d["value"] == pytest.approx(pairs * 10 / 0.05)
        assert d["multi_gpu"]["gathered"]["ms_per_step"] == pytest.approx(5.0)       # both figures, always
        assert d["multi_gpu"]["transport"].startswith("value = the gathered") and d["value_transport"] == "gather"
        if ps is not None:
            assert d["multi_gpu"]["peer_store"] == ps
        else:
            assert d["multi_gpu"]["peer_store"]["available"] is False
    # one transport for a whole curve: --transport gather never takes the peer figure, --transport peer takes it even
    # when it is the slower one (and says so when it is missing)
    args.transport = "gather"
    d = line(good)
    assert d["value_transport"] == "gather" and d["ms_per_step"] == pytest.approx(5.0)
    assert d["multi_gpu"]["peer_store"] == good
    args.transport = "peer"
    d = line(dict(good, ms_per_step=7.0, pairs_per_s=pairs / 7e-3))
    assert d["value_transport"] == "peer_store" and d["ms_per_step"] == 7.0
    d = line({"available": False, "why": "x"})
    assert d["value_transport"] == "gather" and "not available" in d["value_note"]
