"""Batches in which nothing survives the first stage, and empty batches, through every composed path (two batch objects each): the second
object is made from zero images and every file still gets its own answer."""
from _util import emul_api, package


def test_all_failed_and_empty_batches():
    api, pkg = emul_api(), package()
    bad_png = b"\x89PNG\r\n\x1a\n" + b"\0" * 40
    bad_jpg = b"\xff\xd8\xff\xdb" + b"\0" * 30
    outs = api.cs_batch_compress([bad_png, bad_png], pkg.default_parameters(png_optimize=True, width=30))     # PNG resize
    assert [o.code for o in outs] == [30100, 30100]
    assert [o.code for o in api.batch_convert([bad_png, bad_png], pkg.default_parameters(width=30), 3)] == [30100, 30100]   # PNG -> resize -> WebP
    assert [o.code for o in api.batch_convert([bad_png, bad_png], pkg.default_parameters(), 3)] == [30100, 30100]           # PNG -> WebP
    assert [o.code for o in api.batch_convert([bad_png, bad_png], pkg.default_parameters(), 0)] == [30100, 30100]           # PNG -> JPEG
    assert [o.code for o in api.batch_convert([bad_jpg, bad_jpg], pkg.default_parameters(), 1)] == [20100, 20100]           # JPEG -> PNG
    for fmt in (0, 1, 3):
        assert api.batch_convert([], pkg.default_parameters(width=20), fmt) == []
    assert api.cs_batch_compress([], pkg.default_parameters(png_optimize=True, width=30)) == []
