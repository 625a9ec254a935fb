"""The three libm substitutions (sinf / cosf, pow(x, 5), tan) measured at image level on the CPU:
the oracle built with the platform's libm (oracle/liboracle_libm.so) against the oracle with the fixed sequences.
INTEGRATION.md section 5 quotes the figures of tests/libm_sensitivity.py; this test holds them to bounds."""
import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _module():
    spec = importlib.util.spec_from_file_location("libm_sensitivity", os.path.join(ROOT, "tests", "libm_sensitivity.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize("index", [0, 1, 2])
def test_substitutions_do_not_change_the_image(index):
    mod = _module()
    name, (scene, w, h, mb) = list(mod.scenes().items())[index]
    r = mod.measure(scene, w, h, mb, spp=32)
    one, many = r["one_spp"], r["many_spp"]
    # 1 spp: a few pixels differ in the last bits (a sine that is one ulp off); nobody takes another path
    assert one["differing_pixel_fraction"] < 0.05, (name, one)
    assert one["pixels_off_by_more_than_1e-3"] == 0.0, (name, one)
    assert one["max_live_delta_relative"] <= 1e-3, (name, one)      # the stated tolerance on live counts
    # many spp: the difference between the builds is far below the tolerance and far below the noise of either
    assert many["mse_fixed_vs_libm"] < 1e-4, (name, many)
    assert many["mse_fixed_vs_libm"] < 0.01 * many["mse_fixed_vs_fixed_other_iterations"], (name, many)
    assert many["rays_relative_delta"] <= 1e-3, (name, many)
