"""Rasteriser front ends with the reference's interfaces.

``MyRender``          src/READ/gl/myrender.py:12-43  (headless training path; global in src/train.py:596-598)
``Scene``             the camera/cloud state of READ/gl/programs.py::NNScene that the render path reads
``MultiscaleRender``  READ/datasets/dynamic.py:50-99  (viewer / dataset path; GL FBOs replaced by the HIP splat)

Only the ``uv_1d_p1[_dsK]`` input-format tokens (point ids, 1-px points) are rendered — the mode
TexturePipeline uses; other GL modes (colours, normals, splat sizes > 1) are outside the hot path
 and raise NotImplementedError.
"""
import re

import numpy as np
import torch

from . import _lib
from .camera import level_sizes, total_matrix
from .raster import PointCloudRasterizer, index_to_float


def parse_input_string(string):
    """Subset of READ/gl/dataset.py:39-82 needed on this path: mode, point size, downscale."""
    if not re.search('^uv', string):
        raise NotImplementedError(f"input format '{string}': only uv_1d point-id rendering is on the HIP path")
    config = {'mode': 'uv_1d' if 'uv_1d' in string else 'uv_2d'}
    if config['mode'] != 'uv_1d':
        raise NotImplementedError("uv_2d (mesh textures) is outside the point-cloud hot path")
    res = re.findall('ps[0-9]+|p[0-9]+', string)
    config['point_size'] = int(re.search('[0-9]+', res[-1]).group()) if res else 1
    config['splat_mode'] = bool(res) and res[-1].startswith('ps')
    if config['point_size'] != 1 or config['splat_mode']:
        raise NotImplementedError("point sizes > 1 / perspective splats are not on the HIP path")
    res = re.findall('ds[0-5]+', string)
    if res:
        config['downscale'] = int(re.search('[0-9]+', res[-1]).group())
    return config


class MyRender:
    """Same constructor / update_ds / render contract as src/READ/gl/myrender.py.

    ``render(data)`` returns ``(out_dict, depth_dict)``: ``out_dict['id']`` plus one (B,1,h,w) float32
    tensor per input_format token (scale = position in the list, myrender.py:32).  Tensors are CPU
    tensors like the reference's unless ``device_outputs=True`` (then they stay in HBM, int32 ids
    available as ``last_index``)."""

    def __init__(self, ds_list=None, device_outputs=False):
        self.device_outputs = device_outputs
        self.rasterizers = {}
        if ds_list:
            self.update_ds(ds_list)

    def update_ds(self, ds_list):
        self.ds_list = ds_list
        self.ds_ids = [d.id for d in ds_list]
        self.tgt_sh = self.ds_list[0].tgt_sh
        self.rasterizers = {ds.id: PointCloudRasterizer(np.asarray(ds.scene_data['pointcloud']['xyz'], np.float32))
                            for ds in ds_list}

    def render(self, data):
        input_format = self.ds_list[0].input_format.replace(' ', '').split(',')
        ids = data['input']['id']
        ids_t = torch.as_tensor(ids)
        B = len(ids)
        W, H = int(self.tgt_sh[0]), int(self.tgt_sh[1])
        levels = len(input_format)
        proj = data['proj_matrix'].numpy() if torch.is_tensor(data['proj_matrix']) else np.asarray(data['proj_matrix'])
        view = data['view_matrix'].numpy() if torch.is_tensor(data['view_matrix']) else np.asarray(data['view_matrix'])
        tm = total_matrix(proj, view)                                          # myrender.py:28-30
        dev = _lib.require_gpu()
        sizes = level_sizes(W, H, levels)
        index = [torch.zeros((B, h, w), dtype=torch.int32, device=dev) for (w, h) in sizes]
        depth = [torch.zeros((B, h, w), dtype=torch.float32, device=dev) for (w, h) in sizes]
        for ds_id in self.ds_ids:
            sel = torch.where(ids_t == ds_id)[0]
            if sel.numel() == 0:
                continue
            i_l, d_l = self.rasterizers[ds_id].render(tm[sel.numpy()], W, H, levels)
            for l in range(levels):
                index[l][sel.to(dev)] = i_l[l]
                depth[l][sel.to(dev)] = d_l[l]
        self.last_index = index
        out_dict, depth_dict = {'id': ids}, {}
        for l, k in enumerate(input_format):
            f = index_to_float(index[l]).unsqueeze(1)
            d = depth[l].unsqueeze(1)
            out_dict[k] = f if self.device_outputs else f.cpu()
            depth_dict[k] = d if self.device_outputs else d.cpu()
        return out_dict, depth_dict


class Scene:
    """Camera + cloud state with NNScene's setter names (READ/gl/programs.py:330-415), no GL."""

    def __init__(self, xyz=None):
        self.model_matrix = np.eye(4, dtype=np.float32)
        self.view_matrix = np.eye(4, dtype=np.float32)        # camera -> world
        self.proj_matrix = np.eye(4, dtype=np.float32)
        self._raster = None
        self._dirty = True
        self.xyz = None
        if xyz is not None:
            self.set_vertices(xyz)

    def set_vertices(self, positions):
        self.xyz = np.ascontiguousarray(positions, dtype=np.float32)
        self._dirty = True

    def set_model_view(self, m):
        self.model_matrix = np.asarray(m, np.float32)

    def set_camera_view(self, m):
        """m: camera->world pose (viewer.py:264); the GL scene stores inv(m).T, we keep m."""
        self.view_matrix = np.asarray(m, np.float32)

    def set_proj_matrix(self, m):
        self.proj_matrix = np.asarray(m, np.float32)

    def set_use_light(self, use_light):
        pass

    def set_params(self, **kwargs):
        pass

    def rasterizer(self):
        if self._raster is None or self._dirty:
            if self.xyz is None:
                raise ValueError("scene has no point cloud (set_vertices)")
            self._raster = PointCloudRasterizer(self.xyz)
            self._dirty = False
        return self._raster

    def total_matrix(self):
        # clip = P * inv(cam->world) * model * x   (programs.py:121-125 with column vectors)
        view = np.linalg.inv(self.view_matrix.astype(np.float32))
        return (self.proj_matrix @ view @ self.model_matrix).astype(np.float32)[None]


class MultiscaleRender:
    """READ/datasets/dynamic.py:50-99 without OpenGL: one HIP pass fills all five scales; the
    result dict maps each input_format token to an (h, w, 3) float tensor with the point id in
    channel 0 (GL's RGBA32F colour target, programs.py:164-167), row 0 = image top unless gl_frame."""

    def __init__(self, scene, input_format, viewport_size, proj_matrix=None, out_buffer_location='numpy',
                 gl_frame=False, supersampling=1, clear_color=None):
        self.scene = scene
        self.input_format = input_format
        self.proj_matrix = proj_matrix
        self.gl_frame = gl_frame
        self.viewport_size = viewport_size
        self.ss = supersampling
        self.out_buffer_location = out_buffer_location
        self.last_index = None

    def render(self, view_matrix=None, proj_matrix=None, input_format=None):
        if view_matrix is not None:
            self.scene.set_camera_view(view_matrix)
        proj_matrix = self.proj_matrix if proj_matrix is None else proj_matrix
        if proj_matrix is not None:
            self.scene.set_proj_matrix(proj_matrix)
        input_format = input_format if input_format else self.input_format
        fmts = input_format.replace(' ', '').split(',')
        cfgs = [parse_input_string(f) for f in fmts]
        scales = [c.get('downscale', 0) for c in cfgs]
        W, H = self.ss * self.viewport_size[0], self.ss * self.viewport_size[1]
        idx, _ = self.scene.rasterizer().render(self.scene.total_matrix(), W, H, max(scales) + 1, want_depth=False)
        self.last_index = idx
        out = {}
        for fmt, s in zip(fmts, scales):
            ids = index_to_float(idx[s][0])
            if self.gl_frame:
                ids = ids.flip([0])
            x = torch.zeros(ids.shape + (3,), dtype=torch.float32, device=ids.device)
            x[..., 0] = ids
            out[fmt] = x if self.out_buffer_location == 'torch' else x.cpu().numpy()
        return out
