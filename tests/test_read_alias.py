"""CPU: the ``READ/`` alias package merges with another READ tree behind it on sys.path (INTEGRATION.md level 1): render-path
modules come from read_amd, everything else keeps resolving to the other checkout."""
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(code, extra_path):
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([ROOT, extra_path]))
    return subprocess.run([sys.executable, "-c", textwrap.dedent(code)], env=env, capture_output=True, text=True, timeout=300)


def test_alias_package_merges_with_a_tree_behind_it(tmp_path):
    other = tmp_path / "refsrc"
    for d in ("READ", "READ/utils", "READ/datasets", "READ/gl", "READ/pipelines"):
        (other / d).mkdir(parents=True)
        (other / d / "__init__.py").write_text("")
    (other / "READ/utils/perform.py").write_text("class TicToc:\n    origin = 'reference'\n")
    (other / "READ/datasets/dynamic.py").write_text(
        "def get_datasets(args):\n    return 'reference datasets'\nclass MultiscaleRender:\n    origin = 'reference'\n")
    (other / "READ/gl/camera.py").write_text("TRACKBALL = 'reference'\n")
    (other / "READ/pipelines/ogl.py").write_text("class TexturePipeline:\n    origin = 'reference'\n")
    r = _run("""
        import READ, READ.utils.perform, READ.gl.camera, READ.datasets.dynamic as dyn
        from READ.pipelines.ogl import TexturePipeline
        from READ.pipelines import load_pipeline
        from READ.models.unet import UNet
        import read_amd.pipeline, read_amd.render, read_amd.unet
        assert READ.utils.perform.TicToc.origin == 'reference'           # not shadowed by the alias package
        assert READ.gl.camera.TRACKBALL == 'reference'
        assert TexturePipeline is read_amd.pipeline.TexturePipeline      # render path = this repo, not the tree behind
        assert UNet is read_amd.unet.UNet
        assert dyn.MultiscaleRender is read_amd.render.MultiscaleRender  # renderer replaced ...
        assert dyn.get_datasets(None) == 'reference datasets'            # ... the rest of the module kept
        assert len(READ.__path__) >= 2
        print('ok')
    """, str(other))
    assert r.returncode == 0, r.stderr
    assert r.stdout.strip() == "ok"


def test_alias_package_alone_exports_the_renderer():
    r = _run("""
        import READ.datasets.dynamic as dyn, read_amd.render
        assert dyn.MultiscaleRender is read_amd.render.MultiscaleRender and dyn._ref is None
        import importlib
        try:
            importlib.import_module('READ.utils.perform')
        except ModuleNotFoundError:
            print('ok')
    """, "")
    assert r.returncode == 0, r.stderr
    assert r.stdout.strip() == "ok"


def test_alias_package_in_front_of_the_real_reference():
    ref = "/root/reference"
    if not os.path.isdir(os.path.join(ref, "READ")):
        import pytest
        pytest.skip("reference checkout not present")
    r = _run("""
        import READ.utils.perform                                     # a reference-only module that train.py imports
        import READ.pipelines.ogl as o, read_amd.pipeline
        assert o.TexturePipeline is read_amd.pipeline.TexturePipeline
        assert 'reference' in READ.utils.perform.__file__
        print('ok')
    """, ref)
    assert r.returncode == 0, r.stderr


_THIRD_PARTY_STUBS = """
    import sys, types
    # torchvision / cv2 are imported at the top of the reference's dataset modules and are not in this image
    tv, tr = types.ModuleType('torchvision'), types.ModuleType('torchvision.transforms')
    class Compose:
        def __init__(self, ts): self.ts = ts
        def __call__(self, x):
            for t in self.ts: x = t(x)
            return x
    tr.Compose = Compose; tv.transforms = tr
    sys.modules['torchvision'] = tv; sys.modules['torchvision.transforms'] = tr
    sys.modules.setdefault('cv2', types.ModuleType('cv2'))
"""


def test_reference_datasets_use_the_hip_renderer():
    """the reference's DynamicDataset resolves MultiscaleRender / NNScene / app.Window through ITS module globals —
    those must be the HIP-backed classes, and ``get_datasets`` must be importable through the alias (TexturePipeline.create)."""
    ref = "/root/reference"
    if not os.path.isdir(os.path.join(ref, "READ")):
        import pytest
        pytest.skip("reference checkout not present")
    scene = os.path.join(ROOT, "tests", "golden", "scene", "scene.yaml")
    r = _run(_THIRD_PARTY_STUBS + f"""
    import numpy as np
    import READ.datasets.dynamic as dyn, read_amd.render
    from READ.gl.utils import load_scene_data, FastRand, get_proj_matrix
    from READ.gl.programs import NNScene
    from READ.gl.dataset import parse_input_string
    assert dyn._ref is not None, dyn.reference_origin
    assert 'reference' in dyn.reference_origin
    assert callable(dyn.get_datasets)
    DD = dyn.DynamicDataset
    g = DD.__init__.__globals__                                   # what the reference class sees at run time
    assert g['MultiscaleRender'] is read_amd.render.MultiscaleRender
    assert g['NNScene'] is read_amd.render.Scene and NNScene is read_amd.render.Scene
    g['app'].Window(visible=False)                                # "creates GL context": a no-op here
    assert NNScene.MODE_UV == 3 and NNScene.UV_TYPE_1D == 0
    sd = load_scene_data({scene!r})
    n = len(sd['view_matrix'])
    ds = DD(sd, 'uv_1d_p1, uv_1d_p1_ds1', (64, 48), sd['view_matrix'], [''] * n, [''] * n, [''] * n, perturb_points=0.1)
    ds.load()                                                     # NNScene() + setup_scene + FastRand, all without GL
    assert isinstance(ds.scene, read_amd.render.Scene) and ds.scene.xyz.shape[1] == 3
    assert ds.fastrand.toss().shape == (ds.scene.xyz.shape[0], 2)
    K, proj = ds._get_intrinsics()
    assert proj.shape == (4, 4)
    ds.unload()
    assert ds.scene is None
    print('ok')
    """, ref)
    assert r.returncode == 0, r.stderr
