"""Drop-in boundary, checked against the reference's OWN call sites (runs only where /root/reference exists -- this
container; the GPU box has no reference tree, and this container has no GPU, so the calls are bound, not executed):

  * every `droid_backends.<op>(...)` call in networks/modules/corr.py and slam/visual_frontends/visual_frontend.py names
    an op of this project's shim and binds to its signature with the reference's argument count / keywords;
  * the reference's networks/modules/corr.py imports UNMODIFIED against this `droid_backends` (first on sys.path) and its
    autograd Functions reference ops the shim exports;
  * every `pyngp` attribute / method fusion/nerf_fusion.py touches on its live path exists on this project's `pyngp`.
"""
import ast
import importlib.util
import inspect
import os
import sys

import pytest

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not present (GPU box)")


def _calls(path, module_name):
    tree = ast.parse(open(path).read())
    out = []
    for node in ast.walk(tree):
        if isinstance(node, ast.Call) and isinstance(node.func, ast.Attribute) and isinstance(node.func.value, ast.Name) \
                and node.func.value.id == module_name:
            out.append((node.func.attr, len(node.args), [k.arg for k in node.keywords], node.lineno))
    return out


def test_droid_backends_call_sites_bind():
    import droid_backends
    sites = []
    for rel in ("networks/modules/corr.py", "slam/visual_frontends/visual_frontend.py"):
        sites += [(rel,) + c for c in _calls(os.path.join(REF, rel), "droid_backends")]
    assert len(sites) >= 9
    names = {s[1] for s in sites}
    assert {"corr_index_forward", "corr_index_backward", "altcorr_forward", "altcorr_backward", "frame_distance",
            "reduced_camera_matrix", "solve_depth"} <= names
    for rel, name, nargs, kws, line in sites:
        fn = getattr(droid_backends, name, None)
        assert fn is not None, f"{rel}:{line}: droid_backends.{name} missing from the shim"
        try:
            inspect.signature(fn).bind(*([None] * nargs), **{k: None for k in kws})
        except TypeError as e:
            raise AssertionError(f"{rel}:{line}: droid_backends.{name} called with {nargs} args {kws}: {e}")


def test_reference_corr_module_imports_against_the_shim():
    import droid_backends
    spec = importlib.util.spec_from_file_location("ref_corr", os.path.join(REF, "networks/modules/corr.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)                       # `import droid_backends` inside resolves to this project's shim
    assert sys.modules["droid_backends"] is droid_backends and droid_backends.__file__.startswith(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    for cls in ("CorrSampler", "CorrBlock", "CorrLayer", "AltCorrBlock"):
        assert hasattr(mod, cls), cls
    # same constructor / call signatures as this project's host mirror
    from nerfslam import corr as mine
    for cls in ("CorrBlock", "AltCorrBlock"):
        ref_init = list(inspect.signature(getattr(mod, cls).__init__).parameters)
        my_init = list(inspect.signature(getattr(mine, cls).__init__).parameters)
        assert my_init[:len(ref_init)] == ref_init, (cls, ref_init, my_init)
        ref_call = list(inspect.signature(getattr(mod, cls).__call__).parameters)
        my_call = list(inspect.signature(getattr(mine, cls).__call__).parameters)
        assert my_call[:len(ref_call)] == ref_call, (cls, ref_call, my_call)


def test_pyngp_surface_covers_nerf_fusion():
    import pyngp
    src = open(os.path.join(REF, "fusion/nerf_fusion.py")).read()
    tree = ast.parse(src)
    mod_attrs, tb_attrs, train_attrs = set(), set(), set()
    for node in ast.walk(tree):
        if isinstance(node, ast.Attribute):
            v = node.value
            if isinstance(v, ast.Name) and v.id == "ngp":
                mod_attrs.add((node.attr, node.lineno))
            elif isinstance(v, ast.Attribute) and isinstance(v.value, ast.Name) and v.value.id == "self" and v.attr == "ngp":
                tb_attrs.add((node.attr, node.lineno))
            elif isinstance(v, ast.Attribute) and v.attr == "training" and isinstance(v.value, ast.Attribute) and v.value.attr == "nerf":
                train_attrs.add((node.attr, node.lineno))
    # live path of the mapper: construction (:57-101), pose refinement switch (:123), send_data (:285-289), frame (:296-303),
    # evaluation (:388-424); print_ngp_info (:310-340) is a debugging dump of GUI state and is not part of it
    live = lambda line: not (308 <= line <= 345)
    for name, line in sorted(mod_attrs):
        if live(line):
            assert hasattr(pyngp, name), f"nerf_fusion.py:{line}: pyngp.{name}"
    tb = pyngp.Testbed(pyngp.TestbedMode.Nerf, 0)          # no device work before create_empty_nerf_dataset
    for name, line in sorted(tb_attrs):
        if live(line):
            assert hasattr(tb, name), f"nerf_fusion.py:{line}: Testbed.{name}"
    for name, line in sorted(train_attrs):
        if live(line):
            assert hasattr(tb.nerf.training, name), f"nerf_fusion.py:{line}: nerf.training.{name}"
