"""bench.py replays counter-derived figures (HBM traffic, VALU instructions) only from a committed profile of the kernels it is
running AND of the default workload: a profile of another workload (other kernels run there) must never be picked."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_only_default_workload_profiles_of_the_current_kernels_are_replayed(tmp_path):
    import bench
    from eogs2_amd.build import source_hash

    def make(name, sha, args=None):
        d = tmp_path / name
        d.mkdir()
        meta = {"kernel_source_sha256": sha}
        if args is not None:
            meta["bench_args"] = args
        (d / "meta.json").write_text(json.dumps(meta))
        (d / "pmc_mean_per_dispatch.json").write_text("{}")
        return str(d / "pmc_mean_per_dispatch.json")

    cur = source_hash()
    assert bench._profile_is_current(make("r02_v19", cur, ""))
    assert bench._profile_is_current(make("r03_v20", cur))                      # older meta.json without the field
    assert not bench._profile_is_current(make("r02_v19_trained", cur, "--opacity trained"))
    assert not bench._profile_is_current(make("r02_v19_lds", cur, ""))          # suffix: not the traffic passes
    assert not bench._profile_is_current(make("r02_v21", cur, "--size 2048"))   # default name, other workload
    assert not bench._profile_is_current(make("r02_v18", "0" * 64, ""))         # other kernels


def test_committed_default_profile_names_the_kernels_of_the_default_workload():
    """Whatever profile bench.py would replay for the headline line must hold the quad kernels (the default workload's)."""
    import glob

    import bench

    for f in glob.glob(os.path.join(ROOT, "profiles", "r*_v*", "pmc_mean_per_dispatch.json")):
        if not bench._profile_is_current(f):
            continue
        pm = json.load(open(f))
        assert any(k.startswith("render_bwd_quad_kernel") and "FETCH_SIZE" in v and "WRITE_SIZE" in v for k, v in pm.items()), f
