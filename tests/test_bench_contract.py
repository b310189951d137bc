"""CPU: bench.py's one-JSON-line contract on the multi-GPU path, without a GPU (VERDICT r03 next #5c).

The first time bench.py meets an 8-GPU node is the driver's round-end scaling run: nothing of the assembly of that line may
depend on code that has only ever run with one rank.  multi_gpu_fields() builds the multi-GPU part of the line from plain rows;
here it is driven with a faked 8-engine `per_gpu` for every host kind.  And a run that dies must still say where."""
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def _rows(world, n=262144, torch_host=False):
    rows = []
    per = n // world
    for r in range(world):
        lo, hi = r * per, (n if r == world - 1 else (r + 1) * per)          # the reference split, nbody.rs:426-428
        rows.append({"slab": [lo, hi], "force_ms": 1.60 + 0.01 * r, "force_launches": 20, "integrate_ms": 0.004,
                     "bh_eval_ms": 0.0, "bh_eval_launches": 0, "exchange_us": None if torch_host else 30.0 + r, "exchanges": 0 if torch_host else 20,
                     "device_tree_build_ms": 0.0, "device_tree_builds": 0})
    return rows


def test_multi_gpu_fields_for_eight_engines_of_a_group():
    per = _rows(8)
    info = {"exchange": "rccl", "rccl_ranks": 8, "enqueue_threads": 0, "fp32_stale": False, "note": ""}
    out = bench.multi_gpu_fields(per, "group", 8, False, info)
    line = json.loads(json.dumps(out))                                      # serialisable, as the line must be
    assert line["rccl_ranks"] == 8 and line["exchange"] == "rccl" and "exchange_note" not in line
    assert len(line["per_gpu"]) == 8 and line["per_gpu"][7]["slab"] == [229376, 262144]
    assert abs(line["all_gather_us_per_step"] - np.mean([30.0 + r for r in range(8)])) < 1e-12
    assert line["rank_skew"]["kernel_ms_min"] == 1.60 and abs(line["rank_skew"]["kernel_ms_max"] - 1.67) < 1e-12
    assert line["rank_skew"]["exchange_us_max"] == 37.0
    # a group that fell back to peer copies says so
    info = dict(info, exchange="peer_copy_after_rccl_failure", rccl_ranks=0, note="ncclCommInitAll failed: unhandled system error")
    line = json.loads(json.dumps(bench.multi_gpu_fields(per, "group", 8, False, info)))
    assert line["rccl_ranks"] == 0 and line["exchange_note"].startswith("ncclCommInitAll failed")


def test_multi_gpu_fields_for_eight_ranks_of_torch_distributed():
    per = _rows(8, torch_host=True)
    line = json.loads(json.dumps(bench.multi_gpu_fields(per, "torch", 8, False, None, "nccl")))
    assert line["rccl_ranks"] == 8 and line["exchange"] == "torch.distributed nccl"
    assert "all_gather_us_per_step" not in line and line["per_gpu"][0]["exchange_us"] is None
    line = json.loads(json.dumps(bench.multi_gpu_fields(per, "torch", 8, False, None, "gloo")))
    assert line["rccl_ranks"] == 0
    # Barnes-Hut: the skew is that of the traversal
    bh = [dict(r, bh_eval_ms=0.07 + 0.001 * i, force_ms=0.0) for i, r in enumerate(per)]
    line = bench.multi_gpu_fields(bh, "torch", 8, True, None, "nccl")
    assert line["rank_skew"]["kernel_ms_min"] == 0.07


def test_a_run_that_dies_says_where():
    """No GPU here: an 8-engine group cannot be built.  bench.py must exit non-zero, print ONE JSON line on stdout with the stage
    it died in, and the same diagnostics on stderr."""
    if __import__("rust_exp_amd").device_count() > 0:
        import pytest

        pytest.skip("needs a box without a GPU")
    env = dict(os.environ, NBX_GROUP_EXCHANGE="copy")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--n", "4096", "--no-cpu-baseline", "--dry-run"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode == 4, (r.returncode, r.stderr[-400:])
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["value"] is None and d["error"]
    # (engines come up lazily: without a device the first thing that needs one is the first exchange)
    st = d["diagnostics"]["stage"]
    assert st.startswith("host construction") or st.startswith("dry-run: first exchange"), st
    assert d["diagnostics"]["gpus"] == 8 and d["diagnostics"]["host"] == "group"
    assert "FAILED at stage '" + st in r.stderr


def test_companions_block_is_flat_scalars_and_survives_a_failed_child():
    """BASELINE configs #2, #4, #5 ride on the driver-run line. The block is assembled from the children's
    own JSON lines; what goes into `roofline` must be scalars (the driver's parser keeps scalars one level deep); a child that
    fails leaves an `error`, not an exception."""
    assert [k for k, _, _ in bench.COMPANIONS] == ["c0_reference_scene", "c2_brute_65536", "c4_barnes_hut_1048576", "c5_fp16_sources_524288",
                                                   "c6_barnes_hut_2097152"]   # (c6, round 6: config #4's model at twice the size -- hand-overs to the host build)
    argv = {k: a for k, _, a in bench.COMPANIONS}
    assert argv["c4_barnes_hut_1048576"][:6] == ["--workload", "bh", "--bodies", "1048576", "--theta", "0.5"]
    assert "--source-bits" in argv["c5_fp16_sources_524288"] and "524288" in argv["c5_fp16_sources_524288"]
    bh_line = {"value": 1.3e9, "unit": "body-steps/s", "ms_per_step": 0.79, "steps": 40,
               "config": {"tree": "device (bh_build.hip)"},
               "ms_split": {"tree_build": 0.36, "bh_eval_kernel": 0.43},
               "roofline": {"frac": 0.14, "kernel_avg_ms": 0.43, "valu_busy_frac": 0.66, "traffic": 5.9e8,
                            "hbm_algorithmic_bytes_per_launch": 3.7e8, "flops_per_launch": 9.7e9},
               "cpu_baseline": {"value": 9.9e5, "cores": 16, "ms_per_step": 1060.0,
                                "accuracy": {"p999": 1.1e-5, "max": 4e-4, "bodies_beyond_2e-5": 37, "vs": "arbiter"}}}
    c4 = bench.companion_summary("c4_barnes_hut_1048576", bh_line)
    assert c4["build_ms"] == 0.36 and c4["traversal_ms"] == 0.43 and c4["valu_busy"] == 0.66 and c4["cpu_ms_per_step"] == 1060.0
    assert c4["err_p999"] == 1.1e-5 and c4["bodies_beyond_2e-5"] == 37 and c4["frac"] == 0.14
    c2 = bench.companion_summary("c2_brute_65536", {"value": 5.4e12, "unit": "interactions/s", "ms_per_step": 0.79, "steps": 300,
                                                    "roofline": {"frac": 0.6, "kernel_avg_ms": 0.78, "flops_per_interaction": 17,
                                                                 "kernel": "k"}, "cpu_baseline": {"value": 6e9, "cores": 16}})
    flat = bench.flatten_companions({"c2_brute_65536": dict(c2, config="x", argv="y", wall_s=1.0),
                                     "c4_barnes_hut_1048576": dict(c4, config="x", argv="y", wall_s=2.0),
                                     "c5_fp16_sources_524288": {"config": "x", "argv": "y", "error": "rc 1: boom", "wall_s": 0.1}})
    assert flat["c2_value"] == 5.4e12 and flat["c2_frac"] == 0.6 and flat["c4_build_ms"] == 0.36 and flat["c5_error"] == "rc 1: boom"
    assert all(v is None or isinstance(v, (int, float, str)) for v in flat.values())
    # round 6: config #4's model at twice its size rides along as c6_* -- the same summary as c4's, with the steps handed to the host build
    assert argv["c6_barnes_hut_2097152"][:6] == ["--workload", "bh", "--bodies", "2097152", "--theta", "0.5"]
    c6 = bench.companion_summary("c6_barnes_hut_2097152", dict(bh_line, bh_fallbacks=0))
    f6 = bench.flatten_companions({"c6_barnes_hut_2097152": dict(c6, config="x", argv="y", wall_s=9.0)})
    assert f6["c6_fallbacks"] == 0 and f6["c6_build_ms"] == 0.36 and f6["c6_err_p999"] == 1.1e-5 and f6["c6_ms_per_step"] == 0.79
    json.dumps(flat)
    # round 6: the reference's one published number rides on the line as c0_* -- ms per nb_step_barnes_hut call of
    # its default scene, the oracle's 1-thread median beside it (BASELINE.md section 3 row CB), the error against the oracle's traversal
    assert argv["c0_reference_scene"] == ["--workload", "reference_scene"]
    c0_line = {"value": 0.101, "unit": "ms", "steps": 90, "draw_ms": 0.12, "published_ms_per_step": 30.75, "bh_fallbacks": 0,
               "config": {"tree": "device (bh_build.hip), exact sums"},
               "cpu_baseline": {"value": 7.9, "ms_per_step": 7.9, "cores": 1, "accuracy": {"p999": 3e-6, "max": 2e-5, "vs": "orc_bh_forces"}}}
    c0 = bench.companion_summary("c0_reference_scene", c0_line)
    f0 = bench.flatten_companions({"c0_reference_scene": dict(c0, config="x", argv="y", wall_s=3.0)})
    assert f0["c0_ms_per_step"] == 0.101 and f0["c0_cpu_ms_per_step"] == 7.9 and f0["c0_published_ms_per_step"] == 30.75
    assert f0["c0_err_p999"] == 3e-6 and f0["c0_err_max"] == 2e-5 and f0["c0_host_hand_overs"] == 0
    assert all(v is None or isinstance(v, (int, float, str)) for v in f0.values())


def test_cpu_baseline_says_who_ran_it_and_what_it_computes():
    """VERDICT r05 #4 / BASELINE.md section 3 ("state T and CPU model"): the cpu_baseline block names the CPU model, the share of
    the host's logical CPUs the process may use, and the LAW each side computes -- the CPU leg is the 2-D 12-flop reference law
    with an IEEE divide, the GPU sweep the 3-D 17-flop one: the two rates are not the same work per interaction."""
    import numpy as np

    facts = bench.cpu_facts(bench.effective_cores())
    assert isinstance(facts["model"], str) and facts["model"] and facts["logical_cpus"] >= 1
    assert str(bench.effective_cores()) in facts["quota"] and "logical CPUs" in facts["quota"]
    assert "12 flops" in bench.CPU_LAW and "divide" in bench.CPU_LAW and "17" in bench.GPU_LAW_3D and "12" in bench.GPU_LAW_2D
    rng = np.random.default_rng(0)
    st = {k: rng.normal(0, 1, 600).astype(np.float32) for k in ("px", "py", "vx", "vy")}
    st["m"] = np.ones(600, np.float32)
    cb = bench.cpu_baseline(st, 0.05)          # (the oracle on the host cores: the cpu_baseline leg itself, tiny)
    for k in ("value", "unit", "cores", "kind", "sample", "model", "quota", "logical_cpus", "law"):
        assert k in cb, k
    assert cb["kind"] == "port" and cb["cores"] == bench.effective_cores() and cb["law"] == bench.CPU_LAW and cb["value"] > 0
