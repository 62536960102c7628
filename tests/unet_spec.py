"""READ's UNet as (state-dict path, cin, cout, ksize) per BasicConv, written down independently of
the C library so the CPU test-suite can (a) build seeded weights without a GPU
and (b) cross-check the table libreadhip.so exports."""

BASE = 32
UNET_SPEC = []


def _add(path, cin, cout, k):
    UNET_SPEC.append((path, cin, cout, k))


for _n, _P in ((0, BASE * 8), (1, BASE * 4), (2, BASE * 2)):
    _add(f"SCM{_n}.main.0", 8, _P // 4, 3)
    _add(f"SCM{_n}.main.1", _P // 4, _P // 2, 1)
    _add(f"SCM{_n}.main.2", _P // 2, _P // 2, 3)
    _add(f"SCM{_n}.main.3", _P // 2, _P - 8, 1)
    _add(f"SCM{_n}.conv", _P, _P, 1)
for _i, (_ci, _co, _k) in enumerate([(8, 32, 3), (32, 64, 3), (64, 128, 3), (128, 64, 4), (64, 32, 4), (32, 3, 3),
                                     (128, 256, 3), (256, 128, 4)]):
    _add(f"feat_extract.{_i}", _ci, _co, _k)
for _blk, _chs in (("Encoder", [32, 64, 128, 256]), ("Decoder", [256, 128, 64, 32])):
    for _i, _c in enumerate(_chs):
        for _j in range(4):
            _add(f"{_blk}.{_i}.layers.{_j}.main.0", _c, _c, 3)
            _add(f"{_blk}.{_i}.layers.{_j}.main.1", _c, _c, 3)
for _i, (_ci, _co) in enumerate([(256, 128), (128, 64), (64, 32)]):
    _add(f"Convs.{_i}", _ci, _co, 1)
_add("ConvsOut.0", 128, 3, 3)
_add("ConvsOut.1", 64, 3, 3)
for _i in range(3):
    _add(f"AFFs.{_i}.conv.0", 480, 32 << _i, 1)
    _add(f"AFFs.{_i}.conv.1", 32 << _i, 32 << _i, 3)
for _n, _c in ((0, 256), (1, 128), (2, 64)):
    _add(f"FAM{_n}.merge", _c, _c, 3)
