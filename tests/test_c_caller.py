"""The six reference symbols from a plain C program (integration/c-caller/frame_loop.c): prototypes copied from the
Haskell FFI declarations, linked with -lnbody_mi355x like the Rust staticlib was. CPU: it compiles and links against
the library (every symbol resolves). GPU: it runs the experiment's frame loop, twice with the same NB_SEED, and the
framebuffer checksums agree."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "integration", "c-caller", "frame_loop.c")
LIBDIR = os.path.join(ROOT, "rust-exp_amd", "lib")


SRC2 = os.path.join(ROOT, "integration", "c-caller", "baseline_configs.c")


def build(tmp_path, rx, src=SRC):
    rx.lib()   # makes sure the shared library is built
    exe = os.path.join(str(tmp_path), os.path.splitext(os.path.basename(src))[0])
    cc = shutil.which("gcc") or shutil.which("cc")
    assert cc, "no C compiler"
    r = subprocess.run([cc, "-O2", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"), src, "-o", exe, "-L" + LIBDIR,
                        "-lnbody_mi355x", "-Wl,-rpath," + LIBDIR, "-Wl,--no-undefined"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


def test_c_caller_links_against_the_library(tmp_path, rx):
    exe = build(tmp_path, rx)
    out = subprocess.run(["nm", "-D", "--undefined-only", exe], capture_output=True, text=True).stdout
    for sym in ("nb_draw", "nb_step_brute_force", "nb_step_barnes_hut", "nb_random_disk", "nb_stable_orbits", "nb_num_particles"):
        assert sym in out, sym


@pytest.mark.gpu
def test_c_caller_runs_the_frame_loop(tmp_path, rx):
    exe = build(tmp_path, rx)
    env = dict(os.environ, NB_SEED="7")
    outs = []
    for _ in range(2):
        r = subprocess.run([exe, "12"], capture_output=True, text=True, env=env, timeout=300)
        assert r.returncode == 0, r.stdout + r.stderr
        outs.append(r.stdout.strip().splitlines()[-1])
    assert outs[0].startswith("bodies 2000 frames 12")
    assert outs[0].split("lit")[1] == outs[1].split("lit")[1]      # same seed, same pixels
    assert int(outs[0].split("lit")[1].split()[0]) > 1000


def _fnv(st):
    import numpy as np

    h = 1469598103934665603
    words = np.concatenate([np.ascontiguousarray(st[k], dtype="<f4").view(np.uint32) for k in ("px", "py", "pz", "vx", "vy", "vz", "m")])
    # FNV-1a over 32-bit words, 64-bit state (vectorised per word is not possible: fold in Python over a small sample only)
    for w in words.tolist():
        h = ((h ^ w) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h


def test_c_host_generates_the_baseline_workloads(tmp_path, rx):
    """A C host behind the level-2 ABI produces BASELINE.json's inputs itself (nbx_plummer_sphere / nbx_two_galaxies):
    its checksums equal those of the numpy restatement (config #2 and #1 checked here; the golden digests of every size
    are in tests/test_workload_generators.py)."""
    exe = build(tmp_path, rx, SRC2)
    r = subprocess.run([exe, "generate"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = r.stdout.strip().splitlines()
    assert len(lines) == 5 and [int(l.split("bodies")[1].split()[0]) for l in lines] == [1024, 65536, 262144, 1048576, 524288]
    got = {l.split()[0]: int(l.split("fnv")[1].split()[0], 16) for l in lines}
    assert got["#2"] == _fnv(rx.plummer_sphere(65536, dim=3))
    e = rx.NBodyEngine()
    e.seed(1)
    e.stable_orbits(1024, 0.5, 30.0)
    assert got["#1"] == _fnv(e.get_particles())


@pytest.mark.gpu
def test_c_host_runs_the_baseline_configs(tmp_path, rx):
    exe = build(tmp_path, rx, SRC2)
    r = subprocess.run([exe, "run", "3"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = r.stdout.strip().splitlines()
    assert len(lines) == 5 and all("ms/step" in l for l in lines)
    rate3 = float(lines[2].split("ms/step")[1].split()[0])
    assert rate3 > 1e12, lines[2]          # config #3's shape on one GPU: the headline kernel through a C host
