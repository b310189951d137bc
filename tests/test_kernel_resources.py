"""Build-time facts about the bit-exact kernels that a silent compiler change would turn into a large slowdown, checked without a
GPU: no scratch (a spilled summing wave once ran the producer/consumer kernel at 60 % of its speed with identical results) and
VGPR budgets that keep the intended number of waves per SIMD."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "rust-exp_amd", "csrc")


def _metadata(tmp_path, source, extra):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    out = tmp_path / "k.s"
    cmd = [hipcc, "-std=c++17", "-O3", "--offload-arch=gfx950", "-S", "--cuda-device-only", *extra,
           os.path.join(CSRC, source), "-o", str(out)]
    subprocess.run(cmd, check=True, capture_output=True, cwd=CSRC)
    text = out.read_text()
    kernels = {}
    for block in text.split("  - .agpr_count:")[1:]:
        name = re.search(r"\.name:\s+(\S+)", block).group(1)
        kernels[name] = {k: int(re.search(r"\.%s:\s+(\d+)" % k, block).group(1))
                         for k in ("private_segment_fixed_size", "vgpr_count", "sgpr_count", "group_segment_fixed_size",
                                   "vgpr_spill_count", "sgpr_spill_count")}
    return kernels


def _strict_flags():
    mk = open(os.path.join(CSRC, "Makefile")).read()
    strict = re.search(r"^STRICT\s*:=\s*(.*)$", mk, re.M).group(1).split()
    sched = re.search(r"^STRICT_SCHED\s*\?=\s*(.*)$", mk, re.M).group(1).split()
    return strict + sched


def test_bit_exact_all_pairs_kernels_use_no_scratch_and_keep_their_occupancy(tmp_path):
    k = _metadata(tmp_path, "force_strict.hip", _strict_flags())
    pc16 = next(v for n, v in k.items() if "k_force_strict_pcILi16ELi4" in n)
    pc8 = next(v for n, v in k.items() if "k_force_strict_pcILi8ELi8" in n)
    one = next(v for n, v in k.items() if "k_force_strictE" in n)
    for v in (pc16, pc8, one):
        assert v["private_segment_fixed_size"] == 0 and v["vgpr_spill_count"] == 0 and v["sgpr_spill_count"] == 0, v
    assert pc16["vgpr_count"] <= 128 and pc16["group_segment_fixed_size"] == 2 * 30 * 64 * 16      # 16 waves on one CU
    assert pc8["vgpr_count"] <= 128 and pc8["group_segment_fixed_size"] == 2 * 28 * 64 * 16        # two workgroups per CU
    assert 2 * pc8["group_segment_fixed_size"] <= 160 * 1024
    assert one["vgpr_count"] <= 128                                                                 # >= 4 waves per SIMD


def test_default_fast_kernels_keep_eight_waves_per_simd(tmp_path):
    """K1's default all-pairs kernels (variants 6 / 7, both dimensions; K4 runs the same four on the widened fp16 copy) and the
    shared Barnes-Hut walk: no scratch and at most 64 VGPRs, i.e. the 8 waves per SIMD their latency hiding was measured with."""
    k = _metadata(tmp_path, "force_tile.hip", [])
    pkw = [v for n, v in k.items() if "k_force_smem_pkw" in n]
    assert len(pkw) == 4
    for v in pkw:
        assert v["private_segment_fixed_size"] == 0 and v["vgpr_spill_count"] == 0 and v["vgpr_count"] <= 64, v
    mk = open(os.path.join(CSRC, "Makefile")).read()
    strict = re.search(r"^STRICT\s*:=\s*(.*)$", mk, re.M).group(1).split()
    k = _metadata(tmp_path, "bh_eval.hip", strict)
    walks = [v for n, v in k.items() if "k_bh_eval_fast_waveI" in n]       # (not the opt-in k_bh_eval_fast_wave16 A/B kernels)
    assert len(walks) == 5                                                 # 64, 32, 16, 8, 4 bodies per wave
    for v in walks:
        assert v["vgpr_spill_count"] == 0 and v["vgpr_count"] <= 64, v


def test_fold_kernels_of_the_device_tree_build(tmp_path):
    """k_fold_big / k_fold_root (reference fold of the device-built tree): two waves per workgroup, no register spills (a few
    bytes of scratch hold the chunk bookkeeping the compiler indexes dynamically), LDS small enough for eight workgroups per
    CU (every queued node gets its own pair of waves at once)."""
    mk = open(os.path.join(CSRC, "Makefile")).read()
    strict = re.search(r"^STRICT\s*:=\s*(.*)$", mk, re.M).group(1).split()
    k = _metadata(tmp_path, "bh_fold.hip", strict)
    folds = [v for n, v in k.items() if "k_fold_big" in n or "k_fold_root" in n]
    assert len(folds) == 2
    for v in folds:
        assert v["private_segment_fixed_size"] <= 16 and v["vgpr_spill_count"] == 0 and v["sgpr_spill_count"] == 0, v
        assert 8 * v["group_segment_fixed_size"] <= 160 * 1024 and v["vgpr_count"] <= 256, v


def test_cluster_replay_kernels_of_the_device_tree_build(tmp_path):
    """k_cells / k_blobs / k_place (reference fold: EPS clusters replayed on the device).  k_blobs is the one every body runs
    through with the whole replay inlined behind a branch almost nobody takes: it must not spill, must keep its per-wave replay
    record in LDS (four records per workgroup) and leave at least four waves per SIMD; out of line the replay cost the kernel
    191 VGPRs and a scratch frame (measured slower)."""
    mk = open(os.path.join(CSRC, "Makefile")).read()
    strict = re.search(r"^STRICT\s*:=\s*(.*)$", mk, re.M).group(1).split()
    k = _metadata(tmp_path, "bh_cluster.hip", strict)
    blobs = next(v for n, v in k.items() if "k_blobs" in n)
    # (a few SGPRs spill into VGPR lanes -- no memory traffic; nothing goes to scratch beyond the lambda bookkeeping)
    assert blobs["vgpr_spill_count"] == 0 and blobs["sgpr_spill_count"] <= 32 and blobs["private_segment_fixed_size"] <= 16, blobs
    assert blobs["vgpr_count"] <= 128 and 3000 * 4 <= blobs["group_segment_fixed_size"] <= 16 * 1024, blobs
    for name in ("k_cells", "k_place"):
        v = next(v for n, v in k.items() if name in n)
        assert v["private_segment_fixed_size"] == 0 and v["vgpr_spill_count"] == 0 and v["vgpr_count"] <= 64, (name, v)


def test_hand_scheduled_walk_keeps_eight_waves_per_simd(tmp_path):
    """k_bh_walk_groups<*, true> (round 4): the loop owns s20-s56, everything else that lives across it -- kernel arguments used
    after the walk -- comes on top.  80 SGPRs + the 16 the trap handler reserves = 96, the last allocation with eight waves per
    SIMD (800 per SIMD); 82 meant seven: 16 384 walks then take three rounds instead of two and the traversal at a million
    bodies went from 0.430 to 0.465 ms (measured when the folded kick-drift first brought two more pointers along)."""
    mk = open(os.path.join(CSRC, "Makefile")).read()
    strict = re.search(r"^STRICT\s*:=\s*(.*)$", mk, re.M).group(1).split()
    k = _metadata(tmp_path, "bh_walk.hip", strict + ["-Wno-inline-asm"])
    walks = {n: v for n, v in k.items() if "k_bh_walk_groupsILi" in n}
    assert len(walks) == 42                # 64 ... 1 bodies per walk x compiled / hand-scheduled / hand-scheduled with the next child's
                                           # distance overlapped (round 6: the default) x with / without the timeline
    for n, v in walks.items():
        assert v["private_segment_fixed_size"] == 0 and v["vgpr_spill_count"] == 0 and v["sgpr_spill_count"] == 0, (n, v)
        assert v["vgpr_count"] <= 64, (n, v)
        if re.search(r"ILi\d+ELi[12]ELb[01]E", n):                         # <BPW, ASM = 1 | 2, TRACE>: the hand-scheduled loops
            assert v["sgpr_count"] <= 80, (n, v)


def test_warm_sort_kernels_of_the_device_tree_build(tmp_path):
    """bh_sort.hip (round 5): no scratch anywhere; k_bucket_sort within the default 64 KB of dynamic LDS (no per-device opt-in) and
    at most 128 VGPRs (four waves per SIMD); the scatter with its four interleaved descents and records in at most 80."""
    mk = open(os.path.join(CSRC, "Makefile")).read()
    strict = re.search(r"^STRICT\s*:=\s*(.*)$", mk, re.M).group(1).split()
    k = _metadata(tmp_path, "bh_sort.hip", strict)
    names = ("k_sample_rank", "k_keys_scatter", "k_bucket_sort")
    for name in names:
        vs = [v for n, v in k.items() if name in n]
        assert vs, name
        for v in vs:
            assert v["private_segment_fixed_size"] == 0 and v["vgpr_spill_count"] == 0 and v["sgpr_spill_count"] == 0, (name, v)
    assert next(v for n, v in k.items() if "k_bucket_sort" in n)["vgpr_count"] <= 128
    assert all(v["vgpr_count"] <= 80 for n, v in k.items() if "k_keys_scatter" in n)   # (four bodies per thread with their records: six waves per SIMD)
