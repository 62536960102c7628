"""Pipeline plug-in seam of READ (READ/pipelines/pipeline.py:10-71, READ/pipelines/ogl.py:58-154).

``TexturePipeline`` keeps the reference's method/attribute set (export_args, create, state_objects,
dataset_load/unload, extra_optimizer, get_net; model, net, textures, ds_train, ds_val, optimizer,
criterion, args) so ``--pipeline read_amd.pipeline.TexturePipeline`` (or the ``READ.pipelines.ogl``
alias in the ``READ/`` shim package) is selectable by dotted path without editing train.py.
Checkpoints use the reference's format ``{'state_dict', 'args'}`` (READ/utils/train.py:42-66).

Datasets, the VGG criterion and the training loop are NOT rebuilt here (out of the hot path):
``create`` in training mode obtains the datasets exactly as the reference does — ``READ.datasets.dynamic.get_datasets(args)``
(READ/pipelines/ogl.py:6,82; resolved through the ``READ`` alias package, which forwards to the reference checkout behind it on
``sys.path``) unless the caller supplies ``args.get_datasets`` — and the criterion from ``args.criterion_module``.  The training
STEP itself is native: ``UNet.forward`` builds an autograd graph of HIP nodes (read_amd/train.py) and the descriptor optimizer
is ``SparseDescriptorRMSprop``.
"""
import importlib
import os
from pathlib import Path
from types import SimpleNamespace

import torch
from torch import optim

from .net_texture import NetAndTexture
from .texture import PointTexture
from .unet import UNet

TextureOptimizerClass = optim.RMSprop


def _locate(dotted):
    mod, _, name = dotted.rpartition('.')
    return getattr(importlib.import_module(mod), name)


def load_model_checkpoint(path, model):
    """READ/utils/train.py:60-66."""
    ckpt = torch.load(path, map_location='cpu', weights_only=False)
    model.load_state_dict(ckpt['state_dict'])
    return model


def deval_args(args):
    """READ/utils/arguments.py: checkpoints carry plain data only — callables / classes / modules become dotted names."""
    d = dict(vars(args)) if not isinstance(args, dict) else dict(args)
    out = {}
    for k, v in d.items():
        if isinstance(v, (str, int, float, bool, type(None), list, tuple, dict, Path)):
            out[k] = v
        elif hasattr(v, '__module__') and hasattr(v, '__qualname__'):
            out[k] = f"{v.__module__}.{v.__qualname__}"
        else:
            out[k] = repr(v)
    return out


def save_model(save_path, model, args=None):
    """READ/utils/train.py:42-57: {'state_dict', 'args'}."""
    model = model.module if hasattr(model, 'module') else model
    d = {'state_dict': model.state_dict()}
    if args is not None:
        d['args'] = deval_args(args)
    torch.save(d, save_path)


class Pipeline:
    def export_args(self, parser):
        raise NotImplementedError()

    def create(self, args):
        raise NotImplementedError()

    def dataset_load(self, *args, **kwargs):
        pass

    def dataset_unload(self, *args, **kwargs):
        pass

    def get_net(self):
        raise NotImplementedError()

    def extra_optimizer(self, *args):
        return None


class _DeviceAdam(optim.Adam):
    """torch.optim.Adam (READ/pipelines/ogl.py:99) that switches to torch's fused implementation at its first step when every
    parameter lives on the GPU by then.  The pipeline creates its optimizer before `model.cuda()` (as the reference does), where
    `fused=True` is refused; the per-step Python of the multi-tensor implementation over the UNet's 594 tensors was 15 ms of a
    60 ms training step.  Same update rule, same hyper-parameters, same state_dict.
    The fused kernel writes the parameters WITHOUT advancing their version counters (seen on torch 2.10: `_version` stays put,
    the foreach implementation bumps it) — every cache keyed on `_version` (the packed fragments of read_amd/train.py, the
    inference plan's weights key) would go stale, so the stepped parameters are bumped here."""

    def step(self, closure=None):
        if not self.state:                                        # nothing has been stepped yet: the state is created below
            # what Adam.__init__ checks and sets for fused=True (torch/optim/adam.py), done here because the parameters only
            # reach the GPU after the optimizer exists: floating-point parameters on a device the fused kernel supports, not
            # differentiable, no foreach at the same time; and the flag the AMP grad scaler looks for
            params = [p for g in self.param_groups for p in g['params']]
            fusable = (all(p.is_cuda and torch.is_floating_point(p) for p in params)
                       and not any(g.get('differentiable') for g in self.param_groups))
            for g in self.param_groups:
                if g.get('fused') is None and not g.get('foreach'):
                    g['fused'] = True if fusable else None
            if any(g.get('fused') for g in self.param_groups):
                self._step_supports_amp_scaling = True
        stepped = [p for g in self.param_groups if g.get('fused') for p in g['params'] if p.grad is not None]
        out = super().step(closure)
        if stepped:
            inc = getattr(torch._C, '_increment_version', None)
            if inc is not None:
                inc(stepped)
            else:
                with torch.no_grad():
                    for p in stepped:
                        p.add_(0)
        return out


class TexturePipeline(Pipeline):
    def export_args(self, parser):
        add = getattr(parser, 'add', parser.add_argument)
        parser.add_argument('--descriptor_size', type=int, default=8)
        parser.add_argument('--texture_size', type=int)
        parser.add_argument('--texture_ckpt', type=Path)
        add('--texture_lr', type=float, default=1e-1)
        add('--texture_activation', type=str, default='none')
        add('--n_points', type=int, default=0, help='this is for inference')

    @staticmethod
    def _texture(args, size, texture_ckpt=None):
        """``get_texture`` of READ/pipelines/ogl.py:30-42; the src tree passes the checkpoint per dataset
        (src/READ/pipelines/ogl.py:38-48) instead of reading ``args.texture_ckpt``."""
        if getattr(args, 'use_mesh', False):
            raise NotImplementedError("MeshTexture (use_mesh) is outside the point-cloud render path")
        tex = PointTexture(args.descriptor_size, size, activation=getattr(args, 'texture_activation', 'none'),
                           reg_weight=getattr(args, 'reg_weight', 0.))
        ckpt = texture_ckpt if texture_ckpt is not None else getattr(args, 'texture_ckpt', None)
        if ckpt:
            tex = load_model_checkpoint(ckpt, tex)
        return tex

    def create(self, args):
        if isinstance(args, dict):
            args = SimpleNamespace(**args)
        if not hasattr(args, 'descriptor_size'):
            args.descriptor_size = 8
        net = UNet(num_input_channels=8, num_output_channels=3, feature_scale=4, num_res=4)   # ogl.py:19-27
        textures = {}
        self.ds_train = self.ds_val = None
        self.optimizer = self.criterion = self._extra_optimizer = None
        if getattr(args, 'inference', False):
            textures = {0: self._texture(args, int(args.n_points))}
        else:
            get_datasets = getattr(args, 'get_datasets', None)
            if get_datasets is None:                       # READ/pipelines/ogl.py:6,82
                try:
                    get_datasets = importlib.import_module('READ.datasets.dynamic').get_datasets
                except (ImportError, AttributeError) as e:
                    raise RuntimeError("training mode needs READ.datasets.dynamic.get_datasets (put the reference checkout "
                                       "behind this repo on PYTHONPATH, INTEGRATION.md) or args.get_datasets(args)") from e
            got = get_datasets(args)
            # root tree: (ds_train, ds_val) (READ/pipelines/ogl.py:86); src tree: + {ds.id: texture checkpoint or None}
            # (src/READ/pipelines/ogl.py:94,104)
            self.ds_train, self.ds_val = got[0], got[1]
            self.texture_ckpts = dict(got[2]) if len(got) > 2 and got[2] else {}
            for ds in self.ds_train:
                assert ds.scene_data['pointcloud'] is not None, 'set pointcloud'
                textures[ds.id] = self._texture(args, ds.scene_data['pointcloud']['xyz'].shape[0],
                                                self.texture_ckpts.get(ds.id))
            self.optimizer = _DeviceAdam(net.parameters(), lr=args.lr)
            self.sparse_textures = bool(getattr(args, 'sparse_texture_optimizer', True)) and not getattr(args, 'reg_weight', 0.)
            for tex in textures.values():
                tex.sparse_training = self.sparse_textures
            if len(textures) == 1:
                self._extra_optimizer = self._texture_optimizer([next(iter(textures.values()))], args.texture_lr)
            crit = getattr(args, 'criterion_module', None)
            if crit is not None:
                crit = _locate(crit) if isinstance(crit, str) else crit
                self.criterion = crit(**getattr(args, 'criterion_args', {})).cuda()
        self.net = net
        self.textures = textures
        self.model = NetAndTexture(net, textures, getattr(args, 'supersampling', 1))
        self.args = args

    def _texture_optimizer(self, texs, lr):
        """RMSprop over the descriptors (ogl.py:16,99-100): the sparse HIP optimizer unless it is switched off
        (``args.sparse_texture_optimizer = False``) or a dense regulariser (``reg_weight``) needs dense gradients."""
        if getattr(self, 'sparse_textures', False):
            from .train import SparseDescriptorRMSprop
            return SparseDescriptorRMSprop(texs, lr=lr)
        return TextureOptimizerClass([{'params': t.parameters()} for t in texs], lr=lr)

    def state_objects(self):
        objs = {'net': self.net}
        objs.update({ds.name: self.textures[ds.id] for ds in (self.ds_train or [])})
        return objs

    def dataset_load(self, dataset):
        self.model.load_textures([ds.id for ds in dataset])
        for ds in dataset:
            ds.load()

    def extra_optimizer(self, dataset):
        lr_drop = self.optimizer.param_groups[0]['lr'] / self.args.lr
        if self._extra_optimizer is not None:       # single dataset: keep optimizer state across epochs
            self._extra_optimizer.param_groups[0]['lr'] = self.args.texture_lr * lr_drop
            return self._extra_optimizer
        return self._texture_optimizer([self.textures[ds.id] for ds in dataset], self.args.texture_lr * lr_drop)

    def dataset_unload(self, dataset):
        self.model.unload_textures()
        for ds in dataset:
            ds.unload()
            self.textures[ds.id].null_grad()

    def get_net(self):
        return self.net


def load_pipeline(checkpoint, args_to_update=None):
    """READ/pipelines/pipeline.py:34-56: rebuild the pipeline from ckpt['args'], then load the net weights."""
    ckpt = torch.load(checkpoint, map_location='cpu', weights_only=False)
    assert 'args' in ckpt
    a = dict(ckpt['args'])
    if args_to_update:
        a.update(args_to_update)
    # READ/pipelines/pipeline.py:34-56: the checkpoint names its pipeline class by dotted path
    name = a.get('pipeline') or 'READ.pipelines.ogl.TexturePipeline'
    try:
        cls = _locate(name) if isinstance(name, str) else name
    except (ImportError, AttributeError) as e:
        raise ImportError(f"checkpoint pipeline '{name}' cannot be imported") from e
    if not (isinstance(cls, type) and issubclass(cls, Pipeline)):
        raise TypeError(f"checkpoint pipeline '{name}' is not a read_amd Pipeline (only TexturePipeline is on the HIP path)")
    args = SimpleNamespace(**a)
    pipeline = cls()
    pipeline.create(args)
    load_model_checkpoint(checkpoint, pipeline.get_net())
    return pipeline, args


def save_pipeline(pipeline, save_dir, epoch, stage, args):
    """READ/pipelines/pipeline.py:59-71: one .pth per state object."""
    for name, obj in pipeline.state_objects().items():
        filename = f'{obj.__class__.__name__}_stage_{stage}_epoch_{epoch}'
        if name:
            filename = f"{filename}_{name.replace('/', '_')}"
        save_model(os.path.join(save_dir, filename + '.pth'), obj, args=args)
