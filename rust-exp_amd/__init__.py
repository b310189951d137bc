"""rust-exp_amd: MI355X-native N-body hot path of blitzcode/rust-exp behind the reference's nb_* C ABI.

The product is `lib/libnbody_mi355x.so` (hand-written gfx950 HIP kernels + C++ host, built by
`build()` from `csrc/`).  This Python package is only the host-side mirror of the reference
interface used by tests, bench.py and the multi-GPU driver:

  * `nb_*` module functions  = the six `extern "C"` symbols of rs-src/nbody.rs, exactly as
    hs-src/RustNBodyExperiment.hs:101-106 imports them (process-global state);
  * `NBodyEngine`            = the handle-based level-2 ABI (include/nbody_mi355x.h);
  * `ShardedNBody`           = one-process-per-GPU slab sharding with one all-gather per step.

The directory name contains a hyphen (it follows the reference repo's name), so import it
through the `rust_exp_amd` shim module at the repository root.
There is no CPU fallback anywhere in this package: without the HIP library the import fails,
without a GPU every step raises.
"""
from .engine import (  # noqa: F401
    NBX_ERR_INVALID,
    NBX_ERR_NO_DEVICE,
    NBX_ERR_STATE,
    NBX_ERR_TREE,
    NBX_ERR_TREE_DEPTH,
    NBX_K_BH_EVAL,
    NBX_K_EXCHANGE,
    NBX_K_FORCE,
    NBX_K_INTEGRATE,
    NBX_K_TREE_BUILD,
    NBodyEngine,
    NBodyError,
    NBodyGroup,
    build,
    device_count,
    device_info,
    lib,
    lib_path,
    nb_draw,
    nb_num_particles,
    nb_random_disk,
    nb_stable_orbits,
    nb_step_barnes_hut,
    nb_step_brute_force,
)
from .presets import plummer_sphere, splitmix64_uniform, two_galaxies  # noqa: F401
from .sharded import ShardedNBody, reference_slab  # noqa: F401
from .tolerances import fast_step_tolerances  # noqa: F401
