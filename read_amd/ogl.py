"""Viewer-side glue with the interface of READ/gl/nn.py:76-129 (``OGL``): rasterise the scene's
current camera, look up descriptors, run the net, hand back an ``H x W x 4`` RGBA frame.

``OGL(scene, scene_data, viewport_size, net_ckpt, texture_ckpt, ...)`` loads a pipeline checkpoint
exactly like the reference; ``OGL.from_model(scene, model, input_format, viewport_size)`` wraps an
already-built ``NetAndTexture`` (used when no checkpoint file exists, e.g. synthetic scenes).
``infer()`` keeps every stage on the device: int32 index pyramids -> one gather launch -> one UNet
plan, no ToTensor / host copies (nn.py:115-117)."""
import torch

from . import _lib
from .render import MultiscaleRender, is_point_id_pyramid
from .texture import gather_pyramid


class OGL:
    def __init__(self, scene, scene_data, viewport_size, net_ckpt, texture_ckpt, out_buffer_location='numpy',
                 supersampling=1, gpu=True, clear_color=None, temporal_average=False):
        from .pipeline import load_pipeline
        args_upd = {'inference': True}
        if texture_ckpt:
            args_upd['texture_ckpt'] = texture_ckpt
            if 'pointcloud' in scene_data:
                args_upd['n_points'] = scene_data['pointcloud']['xyz'].shape[0]
        pipeline, args = load_pipeline(net_ckpt, args_to_update=args_upd)
        model = pipeline.model
        model.load_textures(0)
        self._setup(scene, model, args.input_format, viewport_size, out_buffer_location, supersampling, gpu,
                    clear_color, temporal_average)

    @classmethod
    def from_model(cls, scene, model, input_format, viewport_size, out_buffer_location='torch', supersampling=1,
                   temporal_average=False):
        self = cls.__new__(cls)
        self._setup(scene, model, input_format, viewport_size, out_buffer_location, supersampling, True, None,
                    temporal_average)
        return self

    def _setup(self, scene, model, input_format, viewport_size, out_buffer_location, supersampling, gpu, clear_color,
               temporal_average):
        if not gpu:
            raise _lib.ReadHipError("OGL(gpu=False): the render path has no CPU implementation")
        self.gpu = True
        self.model = model.cuda().eval()
        if supersampling > 1:
            self.model.ss = supersampling
        self.model.temporal_average = temporal_average
        factor = 16
        assert viewport_size[0] % 16 == 0, f'set width {factor * (viewport_size[0] // factor)}'
        assert viewport_size[1] % 16 == 0, f'set height {factor * (viewport_size[1] // factor)}'
        self.viewport_size = viewport_size
        self.input_format = input_format
        self.renderer = MultiscaleRender(scene, input_format, viewport_size, out_buffer_location='torch',
                                         supersampling=self.model.ss, clear_color=clear_color)
        # the device-resident fast path of infer() serves exactly the layout TexturePipeline trains with: >= 4 tokens,
        # token i = 1-px point ids at downscale i; anything else goes through the checked dict path
        fmts = input_format.replace(' ', '').split(',')
        self._fast_format = len(fmts) >= 4 and is_point_id_pyramid(input_format)
        self.last_path = None                # 'fast' / 'dict': which branch the last infer() took (asserted by the tests)

    def infer(self, input_dict=None):
        """-> {'output': H x W x 4 float tensor (RGB + alpha 1), 'net_input': list of NCHW feature maps}; a caller-supplied
        ``input_dict`` (it must carry its own 'id') is rendered as it is and echoed under 'input', as the src tree's
        ``OGL.infer(input_dict)`` does (src/READ/gl/nn.py:115-137)."""
        model = self.model
        texture = model._modules[str(model._loaded_textures[0])] if model._loaded_textures else model._modules['0']
        fast = (input_dict is None and not model.temporal_average and self._fast_format
                and not self.renderer.scene.augmented() and hasattr(model.net, 'engine'))
        self.last_path = 'fast' if fast else 'dict'
        with torch.set_grad_enabled(False):
            if fast:
                scene = self.renderer.scene
                W, H = self.viewport_size
                fmts = self.input_format.replace(' ', '').split(',')
                raster = scene.rasterizer()
                if raster.n != texture.texture_.shape[-1]:
                    raise ValueError(f"descriptor table has {texture.texture_.shape[-1]} points, the scene cloud {raster.n}")
                ss = int(model.ss)                               # supersampling: raster at ss x, reduce in the gather
                idx, _ = raster.render(scene.total_matrix(), ss * W, ss * H, len(fmts), want_depth=False,
                                       next_total=scene.take_next_total_matrix())      # Scene.announce_next_camera_view
                feats = gather_pyramid(texture.rows(), idx, texture.activation, ss=ss)
                out = model.net.engine(H, W).forward(feats[0][0], feats[1][0], feats[2][0], feats[3][0], channels=4)
                net_input = [f.permute(0, 3, 1, 2) for f in feats]
            else:
                given = input_dict is not None
                if not given:
                    input_dict = {k: v.permute(2, 0, 1)[None] for k, v in self.renderer.render().items()}
                feed = dict(input_dict)                          # the model removes 'id' from the dict it is handed
                if not given or 'id' not in feed:
                    feed['id'] = 0
                o, net_input = model(feed, return_input=True)
                if isinstance(o, dict):                          # src tree: the net returns {'im_out': image}
                    o = o['im_out']
                o = o[0].detach().permute(1, 2, 0)
                out = torch.cat([o, torch.ones_like(o[:, :, :1])], 2).contiguous()
        res = {'output': out, 'net_input': net_input}
        if input_dict is not None:
            res['input'] = input_dict
        return res
