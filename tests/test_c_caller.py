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


def build(tmp_path, rx):
    rx.lib()   # makes sure the shared library is built
    exe = os.path.join(str(tmp_path), "frame_loop")
    cc = shutil.which("gcc") or shutil.which("cc")
    assert cc, "no C compiler"
    r = subprocess.run([cc, "-O2", "-Wall", "-Werror", SRC, "-o", exe, "-L" + LIBDIR, "-lnbody_mi355x",
                        "-Wl,-rpath," + LIBDIR, "-Wl,--no-undefined"], capture_output=True, text=True)
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
