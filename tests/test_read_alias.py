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


_TRAIN_PY_DRIVER = """
    import sys, types, importlib
    def stub(name, **attrs):
        m = types.ModuleType(name); m.__dict__.update(attrs); sys.modules[name] = m; return m
    # third-party packages train.py imports at its top that are not in this image (no network): inert modules.  Nothing of
    # READ itself is stubbed — READ.utils.*, READ.models.compose, READ.pipelines come through the alias package.
    for pkg in ("cv2", "torchvision", "tensorboardX", "munch", "matplotlib", "tqdm"):
        try:
            importlib.import_module(pkg)
        except ImportError:
            if pkg == "torchvision":
                tv = stub(pkg); tv.transforms = stub(pkg + ".transforms"); tv.utils = stub(pkg + ".utils")
            elif pkg == "matplotlib":
                stub(pkg).cm = stub(pkg + ".cm")
            elif pkg == "tensorboardX":
                stub(pkg, SummaryWriter=object)
            elif pkg == "tqdm":
                stub(pkg, tqdm=lambda it, **k: it)
            else:
                stub(pkg)
    path, last_line, want = sys.argv[1], int(sys.argv[2]), sys.argv[3]
    block = "\\n".join(open(path).read().splitlines()[:last_line])
    assert "from READ.models.compose import ModelAndLoss" in block and "from READ.pipelines import save_pipeline" in block
    g = {"__name__": "train_import_block"}
    exec(compile(block, path, "exec"), g)                          # the reference's own import block, verbatim
    import read_amd.net_texture, read_amd.pipeline, read_amd._alias, read_amd.unet
    assert g["ModelAndLoss"] is read_amd.net_texture.ModelAndLoss
    assert g["save_pipeline"] is read_amd.pipeline.save_pipeline
    assert "reference" in sys.modules[g["TicToc"].__module__].__file__      # READ.utils.* still the reference's
    assert "reference" in sys.modules[g["to_device"].__module__].__file__
    assert read_amd._alias.result_convention() == want, read_amd._alias.result_convention()
    # what train.py does next with these names (train.py:405-420 / src/train.py:571-575): the pipeline by dotted path
    pipeline = g["get_module"]("READ.pipelines.ogl.TexturePipeline")()
    assert isinstance(pipeline, read_amd.pipeline.TexturePipeline)
    # the rest of the aliased modules keeps every name of the reference's
    from READ.models.compose import NetAndTexture, MultiscaleNet, RGBTexture, BoxFilter, GaussianLayer
    assert NetAndTexture is read_amd.net_texture.NetAndTexture
    assert "_reference" in BoxFilter.__module__ and BoxFilter(3, 3)(__import__("torch").zeros(1, 3, 8, 8)).shape == (1, 3, 8, 8)
    from READ.models.texture import PointTexture, MeshTexture
    try:
        MeshTexture(3, 64)
        raise SystemExit("MeshTexture must refuse")
    except NotImplementedError:
        pass
    from READ.gl.dataset import generate_input_string, parse_input_string
    assert generate_input_string(parse_input_string("uv_1d_p1_ds2")) == "uv_1d_p1_ds2"
    from READ.models.unet import UNet
    assert UNet is read_amd.unet.UNet
    print("ok")
"""


def _train_py_block(tree, last_line, convention):
    ref = "/root/reference" + ("/src" if tree == "src" else "")
    if not os.path.isfile(os.path.join(ref, "train.py")):
        import pytest
        pytest.skip("reference checkout not present")
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([ROOT, ref]))
    env.pop("READ_AMD_RESULT", None)
    r = subprocess.run([sys.executable, "-c", textwrap.dedent(_TRAIN_PY_DRIVER), os.path.join(ref, "train.py"), str(last_line),
                        convention], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    assert r.stdout.strip().endswith("ok")


def test_root_train_py_import_block_runs_through_the_alias():
    """/root/reference/train.py:1-28 executed verbatim under PYTHONPATH=repo:reference (INTEGRATION level 1) —
    ``from READ.models.compose import ModelAndLoss`` (:26) used to raise ImportError.  Root tree => tensor results."""
    _train_py_block("root", 28, "tensor")


def test_src_train_py_import_block_runs_through_the_alias():
    """/root/reference/src/train.py:1-33 under PYTHONPATH=repo:reference/src; the src tree behind => {'im_out'} results."""
    _train_py_block("src", 33, "dict")


def test_model_and_loss_equals_both_reference_classes():
    """``ModelAndLoss`` against the reference's two classes (READ/models/compose.py:12-32 root, src/READ/models/compose.py:14-42)
    on the same model, criterion, inputs, mask and label: outputs and every loss entry identical."""
    import importlib.util
    import types
    import pytest
    import torch
    from read_amd.net_texture import ModelAndLoss
    if not os.path.isdir("/root/reference/READ"):
        pytest.skip("reference checkout not present")

    def load(path, name):
        added = [m for m in ("imageio", "cv2") if m not in sys.modules]
        for m in added:
            sys.modules[m] = types.ModuleType(m)
        try:
            spec = importlib.util.spec_from_file_location(name, path)
            mod = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(mod)
        finally:
            for m in added:
                sys.modules.pop(m, None)
        return mod
    root = load("/root/reference/READ/models/compose.py", "_ref_compose_root")
    src = load("/root/reference/src/READ/models/compose.py", "_ref_compose_src")
    torch.manual_seed(3)
    conv, seg = torch.nn.Conv2d(4, 3, 3, padding=1), torch.nn.Conv2d(4, 5, 1)

    class TensorNet(torch.nn.Module):
        def forward(self, x, **kw):
            return conv(x)

    class DictNet(torch.nn.Module):
        def forward(self, x, **kw):
            return {'im_out': conv(x), 'seg_out': seg(x)}
    crit = torch.nn.L1Loss()
    x, target = torch.randn(2, 4, 16, 16), torch.rand(2, 3, 16, 16)
    mask = (torch.rand(2, 1, 16, 16) > 0.3).float()
    label = torch.randint(0, 5, (2, 16, 16))
    for use_mask in (False, True):
        o_r, l_r = root.ModelAndLoss(TensorNet(), crit, use_mask=use_mask)(x, target, mask=mask)
        o, l = ModelAndLoss(TensorNet(), crit, use_mask=use_mask)(x, target, mask=mask)
        assert torch.equal(o, o_r) and torch.equal(l, l_r)
        for lab in (None, label):
            o_r, l_r = src.ModelAndLoss(DictNet(), crit, use_mask=use_mask)(x, target, mask=mask, label=lab)
            o, l = ModelAndLoss(DictNet(), crit, use_mask=use_mask)(x, target, mask=mask, label=lab)
            assert set(o) == set(o_r) and all(torch.equal(o[k], o_r[k]) for k in o)
            assert set(l) == set(l_r) == ({'vgg_loss', 'huber_loss'} | ({'seg_loss'} if lab is not None else set()))
            assert all(torch.equal(l[k], l_r[k]) for k in l), {k: (float(l[k]), float(l_r[k])) for k in l}
    # MultiscaleNet (compose.py:184-212), same check
    from read_amd.net_texture import MultiscaleNet
    ins = lambda: {'id': 0, 'a': torch.ones(1, 2, 8, 8), 'b': torch.zeros(1, 2, 8, 8), 'c': torch.ones(1, 2, 4, 4), 'd': torch.ones(1, 2, 4, 4)}

    class Cat(torch.nn.Module):
        def forward(self, *xs, **kw):
            return sum(float(x.sum()) for x in xs), [tuple(x.shape) for x in xs]
    for ss in (1, 2):
        assert MultiscaleNet(Cat(), 2, ss)(ins()) == root.MultiscaleNet(Cat(), 2, ss)(ins())
