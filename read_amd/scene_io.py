"""Scene and camera ingest of READ without trimesh / cv2.

Mirrors the reference's host-side loaders so a real scene directory (``scene.yaml`` + ``pointcloud.ply`` +
Metashape ``camera.xml``) flows into ``Scene`` / ``OGL`` / ``MultiscaleRender`` unchanged:

* ``load_scene_data(path)``          READ/gl/utils.py:258-353 — same keys, same relative-path rule
* ``import_model3d(path)``           READ/gl/utils.py:396-477 — point-cloud branch (``is_mesh=False``)
* ``intrinsics_from_xml`` / ``extrinsics_from_xml`` / ``extrinsics_from_view_matrix``   utils.py:170-208
* ``setup_scene(scene, data)``       READ/gl/utils.py:214-255 — the subset a point-cloud ``Scene`` has
* ``recalc_proj_matrix_planes``, ``rescale_K``, ``crop_intrinsic_matrix``, ``get_xyz_colors``, ``get_valid_matrices``,
  ``fix_relative_path``               utils.py:109-120,153-167,365-389

The PLY reader is written from the PLY 1.0 format description (ascii, binary_little_endian, binary_big_endian;
scalar and list properties) — the reference delegates to ``trimesh.load`` (un-vendored dependency,
``requirement.sh``), of which only ``vertices``, ``colors`` and the raw ``nx, ny, nz`` columns are consumed.

Mesh / texture entries of a scene file are outside the point-cloud render path: they raise
``NotImplementedError`` instead of being silently dropped.
"""
import os
import xml.etree.ElementTree as ET

import numpy as np
import yaml

_PLY_TYPES = {
    'char': 'i1', 'int8': 'i1', 'uchar': 'u1', 'uint8': 'u1', 'short': 'i2', 'int16': 'i2', 'ushort': 'u2',
    'uint16': 'u2', 'int': 'i4', 'int32': 'i4', 'uint': 'u4', 'uint32': 'u4', 'float': 'f4', 'float32': 'f4',
    'double': 'f8', 'float64': 'f8',
}


class PlyError(ValueError):
    pass


def _ply_header(fh):
    """-> (format, [(element name, count, [(prop name, dtype) | (prop name, (count dtype, item dtype))])])."""
    if fh.readline().strip() != b'ply':
        raise PlyError("not a PLY file (missing 'ply' magic)")
    fmt = None
    elements = []
    while True:
        line = fh.readline()
        if not line:
            raise PlyError("unexpected end of file inside the PLY header")
        tok = line.decode('ascii', 'replace').split()
        if not tok or tok[0] in ('comment', 'obj_info'):
            continue
        if tok[0] == 'format':
            if tok[1] not in ('ascii', 'binary_little_endian', 'binary_big_endian'):
                raise PlyError(f"unknown PLY format {tok[1]!r}")
            fmt = tok[1]
        elif tok[0] == 'element':
            elements.append((tok[1], int(tok[2]), []))
        elif tok[0] == 'property':
            if not elements:
                raise PlyError("property before any element")
            if tok[1] == 'list':
                if tok[2] not in _PLY_TYPES or tok[3] not in _PLY_TYPES:
                    raise PlyError(f"unknown PLY list types {tok[2]} {tok[3]}")
                elements[-1][2].append((tok[4], (_PLY_TYPES[tok[2]], _PLY_TYPES[tok[3]])))
            else:
                if tok[1] not in _PLY_TYPES:
                    raise PlyError(f"unknown PLY scalar type {tok[1]!r}")
                elements[-1][2].append((tok[2], _PLY_TYPES[tok[1]]))
        elif tok[0] == 'end_header':
            break
        else:
            raise PlyError(f"unknown PLY header keyword {tok[0]!r}")
    if fmt is None:
        raise PlyError("PLY header has no format line")
    return fmt, elements


def read_ply(path):
    """{element: {property: ndarray}}; list properties become a list of arrays (or one 2-D array when every
    entry has the same length, e.g. triangle faces)."""
    out = {}
    with open(path, 'rb') as fh:
        fmt, elements = _ply_header(fh)
        if fmt == 'ascii':
            tokens = fh.read().split()
            pos = 0
            for name, count, props in elements:
                cols = {p: [] for p, _ in props}
                for _ in range(count):
                    for p, t in props:
                        if isinstance(t, tuple):
                            n = int(tokens[pos])
                            cols[p].append(np.array(tokens[pos + 1:pos + 1 + n], dtype=np.float64).astype(t[1]))
                            pos += 1 + n
                        else:
                            cols[p].append(tokens[pos])
                            pos += 1
                out[name] = {p: (_stack_lists(cols[p]) if isinstance(t, tuple) else
                                 np.array(cols[p], dtype=np.float64).astype(t)) for p, t in props}
            return out
        end = '<' if fmt == 'binary_little_endian' else '>'
        for name, count, props in elements:
            if all(not isinstance(t, tuple) for _, t in props):
                dt = np.dtype([(p, end + t) for p, t in props])
                raw = fh.read(dt.itemsize * count)
                if len(raw) != dt.itemsize * count:
                    raise PlyError(f"PLY element {name!r} is truncated")
                rec = np.frombuffer(raw, dtype=dt, count=count)
                out[name] = {p: np.ascontiguousarray(rec[p]).astype(t) for p, t in props}
            else:                                   # rows of variable length: walk them
                cols = {p: [] for p, _ in props}
                for _ in range(count):
                    for p, t in props:
                        if isinstance(t, tuple):
                            n = int(np.frombuffer(fh.read(np.dtype(t[0]).itemsize), dtype=end + t[0])[0])
                            item = np.dtype(end + t[1])
                            cols[p].append(np.frombuffer(fh.read(item.itemsize * n), dtype=item).astype(t[1]))
                        else:
                            item = np.dtype(end + t)
                            cols[p].append(np.frombuffer(fh.read(item.itemsize), dtype=item)[0])
                out[name] = {p: (_stack_lists(cols[p]) if isinstance(t, tuple) else np.array(cols[p], dtype=t))
                             for p, t in props}
    return out


def _stack_lists(rows):
    if rows and all(len(r) == len(rows[0]) for r in rows):
        return np.stack(rows)
    return rows


def write_ply(path, xyz, rgb=None, normals=None, fmt='binary_little_endian'):
    """Point-cloud PLY writer (tests and fixtures; the reference has none)."""
    xyz = np.asarray(xyz, np.float32)
    cols = [('x', 'f4', xyz[:, 0]), ('y', 'f4', xyz[:, 1]), ('z', 'f4', xyz[:, 2])]
    if normals is not None:
        normals = np.asarray(normals, np.float32)
        cols += [('nx', 'f4', normals[:, 0]), ('ny', 'f4', normals[:, 1]), ('nz', 'f4', normals[:, 2])]
    if rgb is not None:
        rgb = np.asarray(rgb, np.uint8)
        cols += [('red', 'u1', rgb[:, 0]), ('green', 'u1', rgb[:, 1]), ('blue', 'u1', rgb[:, 2])]
    names = {'f4': 'float', 'u1': 'uchar'}
    head = ['ply', f'format {fmt} 1.0', 'comment written by read_amd.scene_io', f'element vertex {len(xyz)}']
    head += [f'property {names[t]} {n}' for n, t, _ in cols] + ['end_header']
    with open(path, 'wb') as fh:
        fh.write(('\n'.join(head) + '\n').encode('ascii'))
        if fmt == 'ascii':
            for i in range(len(xyz)):
                fh.write((' '.join(repr(float(c[i])) if t == 'f4' else str(int(c[i])) for _, t, c in cols) + '\n').encode())
        else:
            end = '<' if fmt == 'binary_little_endian' else '>'
            rec = np.empty(len(xyz), dtype=[(n, end + t) for n, t, _ in cols])
            for n, _, c in cols:
                rec[n] = c
            fh.write(rec.tobytes())


# ---------------------------------------------------------------------------------------------- model import
def get_xyz_colors(xyz, r=8):
    """Position -> colour: every axis normalised to the cloud's bounding box (semantics of READ/gl/utils.py:385-389; ``r`` is
    unused there too)."""
    xyz = np.asarray(xyz)
    lo = xyz.min(axis=0)
    span = xyz.max(axis=0) - lo
    return np.clip((xyz - lo) / span, 0., 1.).astype(np.float32)


def get_normal_colors(normals):
    """Unit normals [-1, 1] -> colours [0, 1] (READ/gl/utils.py:391-393)."""
    return (0.5 * np.asarray(normals) + 0.5).astype(np.float32)


def import_model3d(model_path, uv_order=None, is_mesh=False):
    """READ/gl/utils.py:396-477, point-cloud branch: the dict ``setup_scene`` / ``DynamicDataset`` consume
    (xyz, rgb in [0,1], normals, uv1d = point ids, uv2d zeros, dummy faces, xyz_c)."""
    if is_mesh:
        raise NotImplementedError("mesh import is outside the point-cloud render path (DESIGN.md §6)")
    if not str(model_path).lower().endswith('.ply'):
        raise NotImplementedError(f"only PLY point clouds are read natively, got {model_path}")
    ply = read_ply(model_path)
    if 'vertex' not in ply:
        raise PlyError(f"{model_path} has no vertex element")
    v = ply['vertex']
    for k in 'xyz':
        if k not in v:
            raise PlyError(f"{model_path}: vertex element has no {k!r} property")
    # trimesh keeps float32 file data as float64 vertices; the reference then hands them to float32 GL buffers
    xyz = np.stack([v['x'], v['y'], v['z']], axis=1).astype(np.float64)
    n_pts = xyz.shape[0]
    model = {'rgb': None, 'normals': None, 'uv2d': None, 'faces': None}
    if all(k in v for k in ('red', 'green', 'blue')):
        model['rgb'] = np.stack([v['red'], v['green'], v['blue']], axis=1) / 255.
    if all(k in v for k in ('nx', 'ny', 'nz')):
        normals = np.zeros((n_pts, 3), dtype=np.float32)
        normals[:, 0], normals[:, 1], normals[:, 2] = v['nx'], v['ny'], v['nz']
        model['normals'] = normals
    model['uv2d'] = np.zeros((n_pts, 2), dtype=np.float32)
    model['xyz'] = xyz
    model['xyz_c'] = get_xyz_colors(xyz)
    model['uv1d'] = np.arange(n_pts)
    if model['rgb'] is None:
        model['rgb'] = np.zeros((n_pts, 3), dtype=np.float32)
    if model['normals'] is None:                    # (the reference zeroes rgb here, utils.py:456-458; kept)
        model['rgb'] = np.zeros((n_pts, 3), dtype=np.float32)
    model['faces'] = np.array([0, 1, 2], dtype=np.uint32)
    return model


# ---------------------------------------------------------------------------------------------- cameras
def recalc_proj_matrix_planes(pm, new_near=.01, new_far=1000.):
    """READ/gl/utils.py:109-120."""
    depth = float(new_far - new_near)
    out = pm.copy()
    out[2, 2] = -(new_far + new_near) / depth
    out[2, 3] = -2 * (new_far * new_near) / depth
    return out


def rescale_K(K_, sx, sy, keep_fov=True):
    """Intrinsics of an image resized by (sx, sy): the principal point always moves with the pixels, the focal lengths only
    when the field of view is to be kept (READ/gl/utils.py:153-160)."""
    scale = np.ones((3, 3), dtype=np.asarray(K_).dtype)
    scale[0, 2], scale[1, 2] = sx, sy
    if keep_fov:
        scale[0, 0], scale[1, 1] = sx, sy
    return np.asarray(K_) * scale


def crop_intrinsic_matrix(K, old_size, new_size):
    """Intrinsics after a centred crop from old_size to new_size (w, h): only the principal point scales
    (READ/gl/utils.py:163-167)."""
    out = np.array(K, copy=True)
    for axis in (0, 1):
        out[axis, 2] = new_size[axis] * K[axis, 2] / old_size[axis]
    return out


def intrinsics_from_xml(xml_file):
    """Metashape calibration -> (K fp32 with the principal point at the image centre, (width, height));
    READ/gl/utils.py:170-187."""
    root = ET.parse(xml_file).getroot()
    calibration = root.find('chunk/sensors/sensor/calibration')
    if calibration is None:
        raise ValueError(f"{xml_file}: no chunk/sensors/sensor/calibration element")
    resolution = calibration.find('resolution')
    width = float(resolution.get('width'))
    height = float(resolution.get('height'))
    f = float(calibration.find('f').text)
    K = np.array([[f, 0, width / 2], [0, f, height / 2], [0, 0, 1]], dtype=np.float32)
    return K, (width, height)


def extrinsics_from_xml(xml_file, verbose=False):
    """camera->world 4x4 per aligned camera in file order, y and z axes flipped to the GL convention
    (``extrinsic[:, 1:3] *= -1``); READ/gl/utils.py:190-208.  A label that occurs twice keeps its first position and its
    last transform (the reference collects them in a dict)."""
    cameras = ET.parse(xml_file).getroot().findall('chunk/cameras')[0].findall('camera')
    by_label = {}
    for cam in cameras:
        node = cam.find('transform')
        if node is None:                            # not aligned by Metashape
            if verbose:
                print('failed to align camera', cam.get('label'))
            continue
        by_label[cam.get('label')] = node.text
    flip = np.array([1.0, -1.0, -1.0, 1.0])
    labels = list(by_label)
    poses = [np.array(by_label[l].split(), dtype=np.float64).reshape(4, 4) * flip for l in labels]
    return poses, labels


def get_valid_matrices(mlist):
    """-> (the matrices without NaN / inf entries, their positions in ``mlist``); contract of READ/gl/utils.py:374-382."""
    keep = [i for i, m in enumerate(mlist) if bool(np.all(np.isfinite(m)))]
    return [mlist[i] for i in keep], keep


def extrinsics_from_view_matrix(path):
    """Text file of stacked 4x4 matrices; non-finite ones are dropped, labels are their indices as strings;
    READ/gl/utils.py:211-218 (src numbering)."""
    vm = np.loadtxt(path).reshape(-1, 4, 4)
    vm, ids = get_valid_matrices(vm)
    return vm, [str(i) for i in ids]


def fix_relative_path(path, config_path):
    """A relative path that does not resolve from the working directory is looked up next to the scene file; anything else
    is returned unchanged (contract of READ/gl/utils.py:365-371)."""
    if os.path.isabs(path) or os.path.exists(path):
        return path
    beside = os.path.join(os.path.dirname(config_path), path)
    return beside if os.path.exists(beside) else path


def load_scene_data(path):
    """Scene yaml -> the ``scene_data`` dict the datasets, ``OGL`` and the viewer consume (READ/gl/utils.py:258-353).
    Keys of the yaml: viewport_size, pointcloud, intrinsic_matrix (.xml = Metashape calibration, else a text matrix),
    proj_matrix, view_matrix (.xml or stacked text matrices), model3d_origin, point_sizes, net_path + ckpt + texture_ckpt."""
    with open(path, 'r') as f:
        config = yaml.safe_load(f)
    for key in ('mesh', 'texture'):
        if config.get(key):
            raise NotImplementedError(f"scene files with a {key} entry are outside the point-cloud render path")

    def entry(key):
        """Resolved path of an optional yaml entry, or None."""
        return fix_relative_path(config[key], path) if key in config else None

    out = {'mesh': None, 'texture': None, 'config': config,
           'pointcloud': import_model3d(entry('pointcloud')) if 'pointcloud' in config else None,
           'intrinsic_matrix': None, 'proj_matrix': None, 'view_matrix': None,
           'camera_labels': None,                   # (the reference leaves it unbound without a view_matrix entry)
           'model3d_origin': np.eye(4), 'point_sizes': None, 'net_ckpt': None, 'tex_ckpt': None}
    k_path = entry('intrinsic_matrix')
    if k_path is not None and k_path.endswith('xml'):
        out['intrinsic_matrix'], size = intrinsics_from_xml(k_path)
        assert tuple(config['viewport_size']) == size, f'calibration width, height: ({size[0]}, {size[1]})'
    elif k_path is not None:
        out['intrinsic_matrix'] = np.loadtxt(k_path)[:3, :3]
    if 'proj_matrix' in config:
        out['proj_matrix'] = recalc_proj_matrix_planes(np.loadtxt(entry('proj_matrix')))
    v_path = entry('view_matrix')
    if v_path is not None:
        reader = extrinsics_from_xml if v_path.endswith('xml') else extrinsics_from_view_matrix
        out['view_matrix'], out['camera_labels'] = reader(v_path)
    if 'model3d_origin' in config:
        out['model3d_origin'] = np.loadtxt(entry('model3d_origin'))
    if 'point_sizes' in config:
        out['point_sizes'] = np.load(entry('point_sizes'))
    config['viewport_size'] = tuple(config['viewport_size'])
    if 'net_path' in config:
        ckpt_dir = os.path.join(config['net_path'], 'checkpoints')
        out['net_ckpt'] = fix_relative_path(os.path.join(ckpt_dir, config['ckpt']), path)
        out['tex_ckpt'] = fix_relative_path(os.path.join(ckpt_dir, config['texture_ckpt']), path)
    return out


def setup_scene(scene, data, use_mesh=False, use_texture=False):
    """READ/gl/utils.py:214-255 for a point-cloud ``read_amd.render.Scene``: positions, colours, normals, projection,
    first camera pose, model matrix (faces have no consumer without mesh rendering; per-point size arrays raise)."""
    if use_mesh or use_texture or data.get('pointcloud') is None:
        raise NotImplementedError("only point-cloud scenes are rendered (DESIGN.md §6)")
    pc = data['pointcloud']
    scene.set_vertices(positions=pc['xyz'], colors=pc.get('rgb'), normals=pc.get('normals'), uv1d=pc.get('uv1d'))
    if data.get('proj_matrix') is not None:
        scene.set_proj_matrix(data['proj_matrix'])
    if data.get('view_matrix') is not None and len(data['view_matrix']) > 0:
        scene.set_camera_view(data['view_matrix'][0])
    scene.set_model_view(data['model3d_origin'])
    if data.get('point_sizes') is not None:
        scene.set_point_sizes(data['point_sizes'])


def load_scene(config_path):
    """READ/gl/utils.py:356-362 (which drops its result; this one returns it)."""
    from .render import Scene
    scene_data = load_scene_data(config_path)
    scene = Scene()
    setup_scene(scene, scene_data)
    return scene, scene_data



# ---------------------------------------------------------------------------------------------- small host-side helpers
class FastRand:
    """A bank of ``bank_size`` pre-drawn random arrays ``tform(rand(*shape))``; ``toss()`` hands out one of them (interface of
    READ/gl/utils.py:40-52 — DynamicDataset draws its per-sample point perturbation from it, dynamic.py:176-179,238-239)."""

    def __init__(self, shape, tform, bank_size):
        self.bank = [tform(np.random.rand(*shape)) for _ in range(bank_size)]

    def toss(self):
        return self.bank[np.random.randint(0, len(self.bank))]


def crop_proj_matrix(pm, old_w, old_h, new_w, new_h):
    """Projection matrix of a centred crop / resize of the viewport, term by term as READ/gl/utils.py:94-106 (which notes
    itself that it is approximate; its [1,2] entry is derived from pm[0,2], kept)."""
    out = np.array(pm, copy=True)
    rx, ry = old_w / new_w, old_h / new_h
    ccx = (new_w / 2) / (old_w / 2)
    ccy = (new_h / 2) / (old_h / 2)
    out[0, 0] = pm[0, 0] * rx
    out[0, 2] = (pm[0, 2] - 1) * rx * ccx + 1
    out[1, 1] = pm[1, 1] * ry
    out[1, 2] = (pm[0, 2] + 1) * ry * ccy - 1
    return out
