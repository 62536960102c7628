"""``NetAndTexture``: the texture-lookup + refinement-net composite that READ's pipelines train and
render with (interface of READ/models/compose.py:84-181).

Contract kept from the reference: constructor ``(net, textures, supersampling=1,
temporal_average=False)``; textures live on the CPU until ``load_textures(ids)`` registers them as
sub-modules named ``str(id)`` (so ``.cuda()``, ``state_dict()`` and optimizers see them);
``forward(inputs)`` consumes a dict ``{'id': ids, <token>: (B,1|3,h,w) index maps, ...}``, REMOVES
``'id'`` from it, processes the batch item by item (each item may use a different texture) and
returns ``(B,3,H,W)`` — plus the last item's network inputs when ``return_input=True``.  A net that follows the reference's
``src`` tree and returns ``{'im_out': image[, 'seg_out': logits]}`` (src/READ/models/unet.py:280) gives a dict of batched
tensors here, as src/READ/models/compose.py:134-192 does.  ``ModelAndLoss`` (compose.py:12-32 / src compose.py:14-42),
``MultiscaleNet`` and ``RGBTexture`` complete the module ``READ.models.compose`` that ``train.py:26`` imports from.

MI355X specifics: an item whose inputs are all ``uv*`` index maps (the only layout
TexturePipeline produces) is gathered at every scale by ONE HIP launch into NHWC feature maps that
the UNet engine consumes without a copy.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F
from .texture import _RangeCheck, bilinear_down, gather_pyramid
from .unet import UNet


def _join_items(results):
    """Per-item net results -> one batched result, in the convention the net used (tensor, or dict of tensors)."""
    if isinstance(results[0], dict):
        out = {'im_out': torch.cat([r['im_out'] for r in results], 0)}
        extra = [k for k in results[0] if k != 'im_out']
        for k in extra:                                        # src compose.py:176-183: seg_out only when every item has one
            if all(k in r for r in results):
                out[k] = torch.cat([r[k] for r in results], 0)
        return out
    return torch.cat(results, 0)


def _as_id_list(ids):
    if torch.is_tensor(ids):
        return ids.cpu().tolist() if ids.dim() else [int(ids)]
    return [ids] if isinstance(ids, int) else list(ids)


class NetAndTexture(nn.Module):
    def __init__(self, net, textures, supersampling=1, temporal_average=False):
        super().__init__()
        self.net = net
        self.ss = supersampling
        self.temporal_average = temporal_average
        self.last_input = None
        if not hasattr(textures, 'items'):
            try:
                textures = dict(textures)
            except TypeError:
                textures = {0: textures}
        self._textures = {tid: tex.cpu() for tid, tex in textures.items()}
        self._loaded_textures = []
        self._range_check = _RangeCheck()

    # ---- texture residency -------------------------------------------------------------------
    def load_textures(self, texture_ids):
        ids = _as_id_list(texture_ids)
        for tid in ids:
            self._modules[str(tid)] = self._textures[tid]
        self._loaded_textures = ids

    def unload_textures(self):
        self.check_ids()                                   # a wrong-texture / out-of-range id must not outlive its epoch unreported
        for tid in self._loaded_textures:
            self._modules.pop(str(tid)).cpu()
        self._loaded_textures = []

    def check_ids(self):
        """Block until every queued point-id range check has landed; raises IndexError if a lookup saw an id >= N.
        Covers the checks the training path queues on the textures themselves (PointTexture._range_check)."""
        self._range_check.flush()
        for tex in self._textures.values():
            check = getattr(tex, 'check_ids', None)
            if check is not None:
                check()

    def reg_loss(self):
        return sum((self._modules[str(tid)].reg_loss() for tid in self._loaded_textures), 0)

    # ---- one batch item ------------------------------------------------------------------------
    def _sample_item(self, texture, item):
        """item: ordered {token: (1,c,h,w)}.  Each 'uv' token opens a scale; non-uv tokens that follow it
        are concatenated in front of its texture sample (compose.py:140-160)."""
        tokens = list(item)
        assert 'uv' in tokens[0], 'first input must be uv'
        only_uv = all('uv' in t for t in tokens)
        needs_grad = torch.is_grad_enabled() and texture.texture_.requires_grad
        if only_uv and not needs_grad:
            ids = [texture._ids(item[t]).to(texture.texture_.device) for t in tokens]
            # out-of-range ids (wrong texture for this scene) raise IndexError without a device->host sync per frame:
            # the check of THIS lookup is queued, the previous ones are polled (texture._RangeCheck); the gather clamps
            self._range_check.poll()
            self._range_check.queue(ids[0], texture.texture_.shape[-1])
            feats = gather_pyramid(texture.rows(), ids, texture.activation, ss=self.ss)     # ss > 1: fused bilinear reduce
            return [f.permute(0, 3, 1, 2) for f in feats]
        scales, extras = [], []
        for t in tokens:
            if 'uv' in t:
                scales.append([texture(item[t])])
                extras = scales[-1]
            else:
                extras.insert(len(extras) - 1, item[t].to(texture.texture_.device))
        out = [torch.cat(parts, 1) if len(parts) > 1 else parts[0] for parts in scales]
        if self.ss > 1:                         # compose.py:162-163, as a HIP node (differentiable)
            out = [bilinear_down(x, self.ss) for x in out]
        return out

    def forward(self, inputs, **kwargs):
        texture_ids = _as_id_list(inputs.pop('id'))            # the caller's dict loses 'id', as in the reference
        frames, net_input = [], None
        if torch.is_grad_enabled() and len(texture_ids) > 1 and not self.temporal_average:
            # training: sample every item (each may use its own texture), then ONE network call for the batch — the HIP
            # training graph stacks the items into a single tall image (read_amd/train.py), so a 256x256 crop does not
            # leave most of the chip idle; the per-item results are the same as item-by-item calls — in .train() too: the
            # reference's item-by-item calls give every BatchNorm layer a batch of ONE item and move its running buffers once
            # per item (compose.py:137-176), which the stacked call reproduces with per-item statistic groups
            if isinstance(self.net, UNet):
                kwargs = dict(kwargs, per_item_statistics=True)
            if len(set(texture_ids)) == 1:
                # one scene for the whole batch (the usual case): one lookup per scale instead of one per item and scale —
                # the same values, a fifth of a training step's host time less (8 items x 5 scales of Python per step)
                net_input = self._sample_item(self._modules[str(texture_ids[0])], inputs)
            else:
                per_item = [self._sample_item(self._modules[str(tid)], {k: v[b][None] for k, v in inputs.items()})
                            for b, tid in enumerate(texture_ids)]
                net_input = [torch.cat([it[l] for it in per_item], 0) for l in range(len(per_item[0]))]
            out = self.net(*net_input, **kwargs)
            return (out, net_input) if kwargs.get('return_input') else out
        for b, tid in enumerate(texture_ids):
            texture = self._modules[str(tid)]
            net_input = self._sample_item(texture, {k: v[b][None] for k, v in inputs.items()})
            if self.temporal_average:
                if self.last_input is not None:
                    net_input = [(cur + prev) / 2 for cur, prev in zip(net_input, self.last_input)]
                self.last_input = list(net_input)
            frames.append(self.net(*net_input, **kwargs))
        out = _join_items(frames)
        return (out, net_input) if kwargs.get('return_input') else out


class ModelAndLoss(nn.Module):
    """``ModelAndLoss(model, loss, use_mask=False)``: model and criterion in one module, so ``nn.DataParallel`` scatters both
    (train.py:136, src/train.py:145).  ``forward(*inputs, target, **kwargs) -> (output, loss)``.

    Root-tree convention (READ/models/compose.py:19-32): the model returns the image, ``loss = criterion(output[* mask], target)``.
    ``src``-tree convention (src/READ/models/compose.py:21-42): the model returns ``{'im_out': image[, 'seg_out': logits]}`` and
    the loss is a dict — ``vgg_loss`` = the criterion, ``huber_loss`` = ``F.huber_loss`` (delta 1, mean) on the (masked) image,
    ``seg_loss`` = cross entropy ignoring class 0 when the output carries ``seg_out`` and a ``label`` was passed.  The
    convention is read off the model's result, so one class serves both trees."""

    def __init__(self, model, loss, use_mask=False):
        super().__init__()
        self.model = model
        self.loss = loss
        self.use_mask = use_mask

    def forward(self, *args, **kwargs):
        *inputs, target = args
        output = self.model(*inputs, **kwargs)
        mask = kwargs.get('mask') if self.use_mask else None
        if not isinstance(output, dict):
            return output, self.loss(output if mask is None else output * mask, target)
        image = output['im_out'] if mask is None else output['im_out'] * mask
        losses = {'vgg_loss': self.loss(image, target), 'huber_loss': self._huber(image, target)}
        if 'seg_out' in output and kwargs.get('label') is not None:
            losses['seg_loss'] = F.cross_entropy(output['seg_out'], kwargs['label'], ignore_index=0)
        return output, losses

    @staticmethod
    def _huber(image, target):
        if image.is_cuda:                                  # value and gradient in one HIP launch (csrc/train.hip)
            from .train import huber_loss
            return huber_loss(image, target)
        return F.huber_loss(image, target)


def _reduce_ss(x, ss):
    return x if ss <= 1 else F.interpolate(x, scale_factor=1. / ss, mode='bilinear')


class MultiscaleNet(nn.Module):
    """Image-to-image wrapper of Pix2PixPipeline (compose.py:184-212): consecutive groups of ``input_modality`` inputs are
    concatenated into one scale each.  Pure torch — no descriptor lookup, nothing of the HIP path."""

    def __init__(self, net, input_modality, supersampling=1):
        super().__init__()
        self.net = net
        self.input_modality = input_modality
        self.ss = supersampling

    def forward(self, inputs, **kwargs):
        inputs.pop('id')
        values = list(inputs.values())
        assert len(values) % self.input_modality == 0
        scales = [_reduce_ss(torch.cat(values[i:i + self.input_modality], 1), self.ss)
                  for i in range(0, len(values), self.input_modality)]
        out = self.net(*scales, **kwargs)
        return (out, scales) if kwargs.get('return_input') else out


class RGBTexture(nn.Module):
    """Mesh-texture lookup of RGBTexturePipeline (compose.py:215-235): the only input is ``uv_2d``."""

    def __init__(self, texture, supersampling=1):
        super().__init__()
        self.texture = texture
        self.ss = supersampling

    def forward(self, inputs, **kwargs):
        inputs.pop('id')
        assert list(inputs) == ['uv_2d'], 'check input format'
        uv = inputs['uv_2d']
        out = self.texture(uv)
        return (out, uv) if kwargs.get('return_input') else out
