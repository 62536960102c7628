"""Rasteriser front ends with the reference's interfaces.

``MyRender``          src/READ/gl/myrender.py:12-43  (headless training path; global in src/train.py:596-598)
``Scene``             the camera/cloud state of READ/gl/programs.py::NNScene that the render path reads
``MultiscaleRender``  READ/datasets/dynamic.py:50-99  (viewer / dataset path; GL FBOs replaced by the HIP splat)

``uv_1d_p1[_dsK]`` tokens (point ids, 1-px points: the layout TexturePipeline trains and renders with) take the single-pass
pyramid rasteriser.  Every other token of the input-format DSL (READ/gl/dataset.py:39-82) that makes sense for a point cloud
— ``pN`` point sizes, ``psN`` perspective splats, ``colors``, ``normals_{m,r,l,d}``, ``xyz``, ``depth``, ``labels`` — plus
the dataset augmentations ``set_point_discard`` / ``set_point_perturb`` (READ/gl/programs.py:347-357) is rendered level by
level through ``read_splat_forward_gl`` (the GL twin restated in oracle/raster.c): the z-buffer decides the winning point
of every pixel, the vertex colour of that point (programs.py:133-181, flat shading) is looked up on the device.  Tokens
that need triangles (no ``p``: mesh rendering, ``uv_2d`` mesh textures) and per-point size arrays raise
NotImplementedError.
"""
import re

import numpy as np
import torch

from . import _lib
from .camera import level_sizes, total_matrix
from .raster import PointCloudRasterizer, index_to_float


MODE_COLOR, MODE_NORMALS, MODE_DEPTH, MODE_UV, MODE_XYZ, MODE_LABEL = 0, 1, 2, 3, 4, 5      # NNScene.MODE_* (programs.py:16-21)
UV_TYPE_1D, UV_TYPE_2D = 0, 1
_NORMALS = ('normals_m', 'normals_r', 'normals_l', 'normals_d')


def parse_input_string(string):
    """The input-format DSL of READ/gl/dataset.py:39-82: ``<what>[_<variant>][_pN|_psN][_dsK]``.
    -> {'mode': (mode0, mode1), 'draw_points', 'flat_color', 'point_size', 'splat_mode'[, 'downscale']}."""
    config = {}
    if re.search('^colors', string):
        config['mode'] = (MODE_COLOR, None)
    elif re.search('^uv', string):
        kinds = re.findall('uv_1d|uv_2d', string)
        if not kinds:
            raise ValueError(string)
        config['mode'] = (MODE_UV, UV_TYPE_1D if kinds[-1] == 'uv_1d' else UV_TYPE_2D)
    elif re.search('^normals', string):
        kinds = re.findall('|'.join(_NORMALS), string)
        if not kinds:
            raise ValueError(string)
        config['mode'] = (MODE_NORMALS, _NORMALS.index(kinds[-1]))
    elif re.search('^xyz', string):
        config['mode'] = (MODE_XYZ, None)
    elif re.search('^depth', string):
        config['mode'] = (MODE_DEPTH, None)
    elif re.search('^labels', string):
        config['mode'] = (MODE_LABEL, None)
    else:
        raise ValueError(string)
    sizes = re.findall('ps[0-9]+|p[0-9]+', string)
    config['draw_points'] = config['flat_color'] = bool(sizes)
    config['point_size'] = int(re.search('[0-9]+', sizes[-1]).group()) if sizes else 1
    config['splat_mode'] = bool(sizes) and sizes[-1].startswith('ps')
    scales = re.findall('ds[0-5]+', string)
    if scales:
        config['downscale'] = int(re.search('[0-9]+', scales[-1]).group())
    return config


_WHAT = {MODE_COLOR: 'colors', MODE_UV: 'uv', MODE_NORMALS: 'normals', MODE_XYZ: 'xyz', MODE_DEPTH: 'depth', MODE_LABEL: 'labels'}
_VARIANT = {MODE_UV: ('_1d', '_2d'), MODE_NORMALS: ('_m', '_r', '_l', '_d')}


def generate_input_string(config):
    """Inverse of ``parse_input_string`` (READ/gl/dataset.py:85-122): a draw configuration -> its token,
    ``<what>[_<variant>][_p<N>|_ps<N>][_ds<K>]``.  An unknown uv type raises ValueError like the reference; an unknown normals
    variant is left out like the reference.  ``MODE_LABEL`` gives ``labels`` (the reference function has no branch for it and
    returns a token that does not parse), so ``parse(generate(c)) == c`` holds for every mode ``parse_input_string`` knows."""
    m0, m1 = config['mode']
    s = _WHAT.get(m0, '')
    if m0 == MODE_UV:
        if m1 not in (UV_TYPE_1D, UV_TYPE_2D):
            raise ValueError
        s += _VARIANT[MODE_UV][m1]
    elif m0 == MODE_NORMALS and m1 in (0, 1, 2, 3):
        s += _VARIANT[MODE_NORMALS][m1]
    if config['draw_points']:
        s += ('_ps' if config['splat_mode'] else '_p') + str(config['point_size'])
    if 'downscale' in config:
        s += f"_ds{config['downscale']}"
    return s


def is_point_id_pyramid(input_format):
    """True when the tokens are exactly ``uv_1d_p1`` at downscale 0, 1, 2, ... — the layout served by ONE pass over the
    cloud (pyramid identity)."""
    try:
        cfgs = [parse_input_string(t) for t in input_format.replace(' ', '').split(',')]
    except (ValueError, NotImplementedError):
        return False
    return all(c['mode'] == (MODE_UV, UV_TYPE_1D) and c['draw_points'] and c['point_size'] == 1 and not c['splat_mode']
               and c.get('downscale', 0) == i for i, c in enumerate(cfgs))


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
        if len(self.ds_ids) == 1 and bool((ids_t == self.ds_ids[0]).all()):
            # one scene (the usual batch): the rasteriser's outputs ARE the batch — no index tensors on the device, no copies
            # (a host -> device copy of a pageable tensor blocks the host until the stream has drained: ten of them per call
            # serialised the training loop's host and device sides)
            index, depth = self.rasterizers[self.ds_ids[0]].render(tm, W, H, levels)
            index, depth = list(index), list(depth)
        else:
            index = [torch.zeros((B, h, w), dtype=torch.int32, device=dev) for (w, h) in sizes]
            depth = [torch.zeros((B, h, w), dtype=torch.float32, device=dev) for (w, h) in sizes]
            for ds_id in self.ds_ids:
                sel = torch.where(ids_t == ds_id)[0]
                if sel.numel() == 0:
                    continue
                i_l, d_l = self.rasterizers[ds_id].render(tm[sel.numpy()], W, H, levels)
                sel_d = sel.to(dev)
                for l in range(levels):
                    index[l][sel_d] = i_l[l]
                    depth[l][sel_d] = d_l[l]
        self.last_index = index
        out_dict, depth_dict = {'id': ids}, {}
        for l, k in enumerate(input_format):
            f = index_to_float(index[l]).unsqueeze(1)
            d = depth[l].unsqueeze(1)
            out_dict[k] = f if self.device_outputs else f.cpu()
            depth_dict[k] = d if self.device_outputs else d.cpu()
        return out_dict, depth_dict


class Scene:
    """Camera + cloud state with NNScene's setter names (READ/gl/programs.py:300-415), no GL: positions and the
    per-point attributes the vertex shader reads, the discard / perturb augmentation buffers, the draw parameters
    ``set_params(**parse_input_string(token))`` sets."""

    # the mode constants callers reach through the class (``NNScene.MODE_UV`` ..., READ/gl/programs.py:61-75)
    MODE_COLOR, MODE_NORMALS, MODE_DEPTH, MODE_UV, MODE_XYZ, MODE_LABEL = (MODE_COLOR, MODE_NORMALS, MODE_DEPTH, MODE_UV,
                                                                            MODE_XYZ, MODE_LABEL)
    NORMALS_MODE_MODEL, NORMALS_MODE_REFLECTION, NORMALS_MODE_LOCAL, NORMALS_MODE_DIRECTION, NORMALS_MODE_RAW = 0, 1, 2, 3, 4
    UV_TYPE_1D, UV_TYPE_2D = UV_TYPE_1D, UV_TYPE_2D

    def __init__(self, xyz=None, flat_color=True):
        self.model_matrix = np.eye(4, dtype=np.float32)
        self.view_matrix = np.eye(4, dtype=np.float32)        # camera -> world
        self.proj_matrix = np.eye(4, dtype=np.float32)
        self._raster = None
        self._dirty = True
        self.xyz = None
        self.colors = self.normals = None
        self._dev = {}
        self.point_discard = None         # bool (N,)  set_point_discard  (programs.py:347-351)
        self.point_perturb = None         # float (N,2) set_point_perturb (programs.py:353-357)
        self.point_sizes = None           # float (N,) set_point_sizes (programs.py:339-345)
        self.point_drop = None            # (p, seed): seeded drop evaluated on the device
        self.point_perturb_seeded = None  # (amp, seed)
        self.params = {'mode': (MODE_UV, UV_TYPE_1D), 'draw_points': True, 'flat_color': True, 'point_size': 1,
                       'splat_mode': False}
        if xyz is not None:
            self.set_vertices(xyz)

    def set_vertices(self, positions, colors=None, normals=None, uv1d=None, uv2d=None, texture=None):
        positions = np.asarray(positions)
        for name, a in (('colors', colors), ('normals', normals), ('uv1d', uv1d), ('uv2d', uv2d)):
            assert a is None or positions.shape[0] == np.asarray(a).shape[0], 'arrays must have the same shape[0]'
        self.xyz = np.ascontiguousarray(positions, dtype=np.float32)
        self.colors = None if colors is None else np.ascontiguousarray(colors, dtype=np.float32)
        self.normals = None if normals is None else np.ascontiguousarray(normals, dtype=np.float32)
        if uv1d is not None and not np.array_equal(np.asarray(uv1d).reshape(-1), np.arange(positions.shape[0])):
            raise NotImplementedError("uv1d other than the point index (import_model3d's arange) is not supported")
        self.xyz_min, self.xyz_max = self.xyz.min(axis=0), self.xyz.max(axis=0)          # programs.py:334-335
        self.point_discard = self.point_perturb = self.point_sizes = None
        self._dev = {}
        self._dirty = True

    def set_point_sizes(self, point_sizes):
        """Per-point sizes (READ/gl/programs.py:339-345, scene yaml 'point_sizes'): from now on every token is drawn with the
        point's own size instead of the token's N — the reference sets global_point_size to 0 and set_params skips
        'point_size' (programs.py:404-406); "ps" tokens still divide by clip z."""
        ps = np.ascontiguousarray(point_sizes, dtype=np.float32).reshape(-1)
        if self.xyz is not None and ps.shape[0] != self.xyz.shape[0]:
            raise ValueError(f"point_sizes has {ps.shape[0]} entries for {self.xyz.shape[0]} points")
        self.point_sizes = ps

    def set_point_discard(self, arr):
        self.point_discard = None if arr is None else np.ascontiguousarray(arr).astype(bool)

    def set_point_perturb(self, arr):
        self.point_perturb = None if arr is None else np.ascontiguousarray(arr, dtype=np.float32).reshape(-1, 2)

    def set_point_drop(self, p, seed=0):
        """Seeded form of ``set_point_discard(np.random.rand(N) < p)`` (READ/datasets/dynamic.py:235-236): evaluated on
        the device from a hash of (point id, seed), restated bit for bit by oracle.drop_mask."""
        self.point_drop = (float(p), int(seed)) if p else None

    def set_point_perturb_seeded(self, amp, seed=0):
        """Seeded form of ``set_point_perturb(amp * (rand(N,2) - 0.5))`` (dynamic.py:176-179,238-239)."""
        self.point_perturb_seeded = (float(amp), int(seed)) if amp else None

    def set_model_view(self, m):
        self.model_matrix = np.asarray(m, np.float32)

    def set_camera_view(self, m):
        """m: camera->world pose (viewer.py:264); the GL scene stores inv(m).T, we keep m."""
        self.view_matrix = np.asarray(m, np.float32)

    def announce_next_camera_view(self, m):
        """Extension (not in NNScene): the pose the NEXT frame will be rendered from, when the caller knows it — a trajectory replay,
        a sweep, a viewer that extrapolates its camera.  The rasteriser then prepares that frame inside this frame's last launch
        (PointCloudRasterizer.render(next_total=...)); a wrong announcement costs two small memsets, never a wrong pixel.
        Consumed by the next render; None withdraws it."""
        self.next_view_matrix = None if m is None else np.asarray(m, np.float32)

    def take_next_total_matrix(self):
        m = getattr(self, 'next_view_matrix', None)
        self.next_view_matrix = None
        if m is None:
            return None
        return (self.proj_matrix @ np.linalg.inv(m.astype(np.float32)) @ self.model_matrix).astype(np.float32)[None]

    def set_proj_matrix(self, m):
        self.proj_matrix = np.asarray(m, np.float32)

    def set_use_light(self, use_light):
        if use_light:
            raise NotImplementedError("the viewer's lighting pass is not part of the render path")

    def set_params(self, skip=(), **kwargs):
        """programs.py:404-416: every key with a setter is applied; here the draw parameters are simply recorded."""
        for k, v in kwargs.items():
            if k not in skip:
                self.params[k] = v

    def delete(self):
        """NNScene.delete() (READ/gl/programs.py; DynamicDataset.unload, dynamic.py:181-183): drop the device copies."""
        self._raster = None
        self._dev = {}
        self._dirty = True

    def augmented(self):
        return (self.point_discard is not None or self.point_perturb is not None or self.point_drop is not None
                or self.point_perturb_seeded is not None or self.point_sizes is not None)

    def device_array(self, name):
        """colors / normals / xyz as (N,3) CUDA tensors, uploaded on first use."""
        if name not in self._dev:
            a = getattr(self, name)
            if a is None:
                a = np.zeros((self.xyz.shape[0], 3), np.float32)        # programs.py:326-327: missing attributes are zeros
            self._dev[name] = torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(_lib.require_gpu())
        return self._dev[name]

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
        self.scene.set_use_light(False)
        input_format = input_format if input_format else self.input_format
        fmts = input_format.replace(' ', '').split(',')
        scene = self.scene
        W, H = self.ss * self.viewport_size[0], self.ss * self.viewport_size[1]
        out = {}
        if is_point_id_pyramid(input_format) and not scene.augmented() and W % (1 << (len(fmts) - 1)) == 0 \
                and H % (1 << (len(fmts) - 1)) == 0:
            # the layout of TexturePipeline: one pass over the cloud feeds every scale
            idx, _ = scene.rasterizer().render(scene.total_matrix(), W, H, len(fmts), want_depth=False)
            self.last_index = idx
            for fmt, ids_l in zip(fmts, idx):
                out[fmt] = self._package(self._id_image(ids_l[0]), fmt)
            return out
        self.last_index = []
        for fmt in fmts:
            cfg = parse_input_string(fmt)
            scene.set_params(**cfg)
            s = cfg.get('downscale', 0)
            w, h = W // 2 ** s, H // 2 ** s                       # dynamic.py:61: ss * viewport // 2**i
            x = self._render_token(cfg, w, h)
            out[fmt] = self._package(x, fmt)
        return out

    # ---- one token = one GL draw of the reference (READ/gl/render.py:52-85) -------------------------------------------
    def _id_image(self, ids):
        x = torch.zeros(ids.shape + (3,), dtype=torch.float32, device=ids.device)
        x[..., 0] = index_to_float(ids)
        return x

    def _package(self, x, fmt):
        if self.gl_frame:
            x = x.flip([0])
        if ('depth' in fmt and 'depth3' not in fmt) or 'label' in fmt:          # dynamic.py:92-95
            x = x[..., :1]
        return x if self.out_buffer_location == 'torch' else x.cpu().numpy()

    def _render_token(self, cfg, w, h):
        scene = self.scene
        if not cfg['draw_points']:
            raise NotImplementedError("tokens without a point size draw triangles (mesh rendering); a point cloud needs pN / psN")
        mode0, mode1 = cfg['mode']
        if mode0 == MODE_UV and mode1 == UV_TYPE_2D:
            raise NotImplementedError("uv_2d (mesh textures) is outside the point-cloud render path")
        M = scene.total_matrix()
        idx, dep = scene.rasterizer().render_gl(M, w, h, point_size=cfg['point_size'], relative=cfg['splat_mode'],
                                                min_point_size=1.0, discard=scene.point_discard, drop=scene.point_drop,
                                                perturb=scene.point_perturb, perturb_hash=scene.point_perturb_seeded,
                                                point_sizes=scene.point_sizes)
        self.last_index.append(idx)
        ids, covered = idx[0], (dep[0] != 0) | (idx[0] != 0)
        if mode0 == MODE_UV:
            return self._id_image(ids)
        lid = ids.long()
        if mode0 == MODE_COLOR:
            col = scene.device_array('colors')[lid]
        elif mode0 == MODE_LABEL:                                                # programs.py:176-178
            col = torch.zeros(ids.shape + (3,), dtype=torch.float32, device=ids.device)
            col[..., 0] = scene.device_array('normals')[lid][..., 0] / 255.
        elif mode0 == MODE_XYZ:                                                  # programs.py:172-175
            lo = torch.from_numpy(scene.xyz_min).to(ids.device)
            hi = torch.from_numpy(scene.xyz_max).to(ids.device)
            col = (scene.device_array('xyz')[lid] - lo) / (hi - lo + 1e-9)
        elif mode0 == MODE_DEPTH:                                                # programs.py:160-164: gl_Position.z
            pos = scene.device_array('xyz')[lid]
            m2 = torch.from_numpy(M[0, 2].copy()).to(ids.device)
            d = m2[0] * pos[..., 0] + m2[1] * pos[..., 1] + m2[2] * pos[..., 2] + m2[3]
            col = d[..., None].expand(-1, -1, 3)
        else:                                                                    # programs.py:137-159
            nrm = scene.device_array('normals')[lid]
            cam = torch.from_numpy(np.ascontiguousarray(scene.view_matrix[:3, 3])).to(ids.device)
            unit = lambda v: v / v.norm(dim=-1, keepdim=True)
            if mode1 == 0:
                col = nrm * 0.5 + 0.5
            else:
                vdir = unit(cam - scene.device_array('xyz')[lid])
                if mode1 == 1:                                                   # reflect(I, N) = I - 2 dot(N, I) N
                    col = unit(vdir - 2.0 * (nrm * vdir).sum(-1, keepdim=True) * nrm) * 0.5 + 0.5
                elif mode1 == 2:                                                 # normal in the camera frame
                    w2c = torch.from_numpy(np.linalg.inv(scene.view_matrix.astype(np.float32))).to(ids.device)
                    p = cam + nrm
                    local = p @ w2c[:3, :3].T + w2c[:3, 3]
                    col = unit(local) * 0.5 + 0.5
                else:
                    col = vdir * 0.5 + 0.5
        return torch.where(covered[..., None], col.float(), torch.zeros((), device=ids.device))
