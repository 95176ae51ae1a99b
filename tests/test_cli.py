"""The caesiumclt shell (caesium-clt_amd/cli): the reference's own CLI tests restated.

  options.rs:264-420      validators            -> test_validators_*, test_flag_*
  scan_files.rs:173-395   scanner + base path   -> test_scan_*, test_base_folder_*
  compressor.rs:615-1060  paths + per-file policy-> test_output_path, test_policy_*
  main.rs:361-760         recap + JSON          -> test_json_*, test_recap_*

CPU tests run `tests/emul/cli_probe` (function-level) and `tests/emul/caesiumclt_emul` (the same cli.cpp linked to the
emulation build of the kernels); the `gpu` test runs the product binary caesium-clt_amd/bin/caesiumclt.
"""
import json
import os
import subprocess
import time

import pytest

from _util import ROOT, emul_api, oracle_lossless, oracle_lossy, oracle_resized
from gen_synth import synth_jpeg

PROBE = os.path.join(ROOT, "tests", "emul", "cli_probe")
EMUL_CLI = os.path.join(ROOT, "tests", "emul", "caesiumclt_emul")
PRODUCT_CLI = os.path.join(ROOT, "caesium-clt_amd", "bin", "caesiumclt")


@pytest.fixture(scope="module", autouse=True)
def _built():
    emul_api()  # (re)builds the emulation targets, incl. the two CLI binaries, when sources changed
    assert os.path.exists(PROBE) and os.path.exists(EMUL_CLI)


def probe(*a):
    return subprocess.run([PROBE, *map(str, a)], capture_output=True, text=True, check=True).stdout.rstrip("\n")


def run_cli(binary, *a, cwd=None):
    return subprocess.run([binary, *map(str, a)], capture_output=True, text=True, cwd=cwd)


def parsed(*flags):
    out = probe("args", *flags)
    if out.startswith("ERR"):
        return None, out[4:]
    return dict(kv.split("=", 1) for kv in out.split()[1:]), None


@pytest.fixture()
def tree(tmp_path):
    """samples/-like tree: j0, level_1_0/j1, level_1_0/level_2_0/j2, a text file and an extension liar"""
    root = tmp_path / "samples"
    (root / "level_1_0" / "level_2_0").mkdir(parents=True)
    files = {"j0.JPG": synth_jpeg(40, 320, 200, texture=25), "level_1_0/j1.jpg": synth_jpeg(41, 200, 320, texture=25),
             "level_1_0/level_2_0/j2.jpeg": synth_jpeg(42, 96, 64, texture=25)}
    for rel, data in files.items():
        (root / rel).write_bytes(data)
    (root / "notes.txt").write_text("x" * 64)
    (root / "fake.jpg").write_text("not an image at all, really")
    return root, files


def test_exif_ifd_offset_near_4gib_does_not_wrap(tmp_path):
    """an APP1 whose IFD offset is 0xFFFFFFFE passed the 32-bit bounds check and read 4 GB past the buffer"""
    src = synth_jpeg(7, 64, 48)
    tiff = b"II*\x00" + (0xFFFFFFFE).to_bytes(4, "little") + b"\x00" * 16
    app1 = b"\xff\xe1" + (2 + 6 + len(tiff)).to_bytes(2, "big") + b"Exif\x00\x00" + tiff
    f = tmp_path / "wrap.jpg"
    f.write_bytes(src[:2] + app1 + src[2:])
    assert probe("dims", f, 1) == "64 48"


# ------------------------------------------------------------------------------------------------ clap short-option forms
def test_short_option_clusters_and_attached_values():
    """clap accepts -RS, -Rd, -q80, -q=80, -Oall, -o=dir, -RSq 80 and a bare '-' positional; each parses to what
    the spelled-out form gives"""
    ref, _ = parsed("-q", "80", "-o", "out", "-R", "-S", "-d", "-O", "all", "x.jpg")
    for form in (("-q80", "-oout", "-RSd", "-Oall", "x.jpg"), ("-q=80", "-o=out", "-RS", "-d", "-O=all", "x.jpg"),
                 ("-RSdq", "80", "-o", "out", "-O", "all", "x.jpg"), ("-dRSq80", "-Oall", "-oout", "x.jpg")):
        got, err = parsed(*form)
        assert err is None and got == ref, (form, got, err)
    got, err = parsed("--lossless", "-o", "out", "-eQ", "-")
    assert err is None and got["exif"] == "1" and got["quiet"] == "1" and got["files"] == "1"
    _, err = parsed("-q80", "-oout", "-RZ", "x.jpg")
    assert err == "unexpected argument '-Z' found"
    got, err = parsed("-q", "80", "-o", "out", "--", "-RS")     # behind '--' everything is a file
    assert err is None and got["recursive"] == "0" and got["files"] == "1"


# ------------------------------------------------------------------------------------------------ validators
def test_validators_max_size():
    want = {"10000": 10000, "1000000": 1000000, "1KB": 1000, "1KiB": 1024, "1MB": 1_000_000, "1MiB": 1_048_576, "0.3GB": 300_000_000, "0.5GiB": 536_870_912}
    for text, n in want.items():
        assert probe("bytesize", text) == str(n), text
    for bad in ("invalid", "1XB", ""):
        assert probe("bytesize", bad) == "ERR"


def test_validators_min_savings():
    for text, v in (("10%", 10.0), ("0%", 0.0), ("100%", 100.0), ("1.5%", 1.5), ("0.1%", 0.1), ("99.9%", 99.9)):
        kind, val = probe("minsavings", text).split()
        assert kind == "pct" and abs(float(val) - v) < 1e-9
    for bad in ("101%", "-1%", "", "abc", "%"):
        assert probe("minsavings", bad).startswith("ERR"), bad
    for text, n in (("100KB", 100_000), ("1MB", 1_000_000), ("1MiB", 1_048_576), ("1B", 1), ("100", 100)):
        assert probe("minsavings", text) == f"bytes {n}"


def test_validators_ranges():
    for q in (0, 50, 100):
        assert parsed("-q", q, "-o", "x", "f")[0]["quality"] == str(q)
    assert "Quality must be between 0 and 100" in parsed("-q", 101, "-o", "x", "f")[1]
    assert parsed("-q", "abc", "-o", "x", "f")[1] == "'abc' is not a valid number"
    assert parsed("-q", 80, "-o", "x", "--verbose", 3, "f")[0]["verbose"] == "3"
    assert "Verbosity must be between 0 and 3" in parsed("-q", 80, "-o", "x", "--verbose", 4, "f")[1]
    assert parsed("-q", 80, "-o", "x", "--png-opt-level", 6, "f")[0]["png"] == "6"
    assert "PNG optimization level must be between 0 and 6" in parsed("-q", 80, "-o", "x", "--png-opt-level", 7, "f")[1]


def test_flag_enums_and_defaults():
    o, _ = parsed("--lossless", "--same-folder-as-input", "a", "b")
    assert o["lossless"] == "1" and o["same"] == "1" and o["files"] == "2" and o["format"] == "5" and o["overwrite"] == "0"
    assert o["verbose"] == "1" and o["threads"] == "0" and o["png"] == "3" and o["chroma"] == "0" and o["gpus"] == "1"
    for text, n in (("4:4:4", 444), ("4:2:2", 422), ("4:2:0", 420), ("4:1:1", 411), ("auto", 0)):
        assert parsed("-q", 1, "-o", "x", "--jpeg-chroma-subsampling", text)[0]["chroma"] == str(n)
    for i, text in enumerate(("all", "never", "bigger")):
        assert parsed("-q", 1, "-o", "x", "-O", text)[0]["overwrite"] == str(i)
    for i, text in enumerate(("jpeg", "png", "gif", "webp", "tiff", "original")):
        assert parsed("-q", 1, "-o", "x", "--format", text)[0]["format"] == str(i)
    assert parsed("-q", 1, "-o", "x", "--format", "bmp")[1].startswith("invalid value")
    o, _ = parsed("--max-size=0.5MB", "--output=out", "-RSd", "x") if False else parsed("--max-size=0.5MB", "--output=out", "-R", "-S", "-d", "x")
    assert o["max_size"] == "500000" and o["output"] == "out" and o["recursive"] == o["keep_structure"] == o["dry"] == "1"


def test_flag_groups():
    assert "required" in parsed("-o", "x", "f")[1]                       # compression group is required
    assert "required" in parsed("-q", 80, "f")[1]                         # output destination group is required
    assert "cannot be used with" in parsed("-q", 80, "--lossless", "-o", "x", "f")[1]
    assert "cannot be used with" in parsed("-q", 80, "--max-size", "1KB", "-o", "x", "f")[1]
    assert "cannot be used with" in parsed("-q", 80, "-o", "x", "--same-folder-as-input", "f")[1]
    assert "cannot be used with" in parsed("-q", 80, "-o", "x", "--width", 10, "--long-edge", 10, "f")[1]
    assert "cannot be used with" in parsed("-q", 80, "-o", "x", "--long-edge", 10, "--short-edge", 10, "f")[1]
    assert "cannot be used together" in parsed("-q", 80, "-o", "x", "--quiet", "--json", "f")[1]
    assert "unexpected argument" in parsed("-q", 80, "-o", "x", "--frobnicate", "f")[1]
    assert parsed("-q", 80, "-o", "x", "--width", 10, "--height", 20, "f")[0]["height"] == "20"


def test_parallelism_count():
    # main.rs:361-379
    assert probe("threads", 4, 8) == "4" and probe("threads", 0, 8) == "8" and probe("threads", 16, 8) == "8" and probe("threads", 1, 1) == "1"


# ------------------------------------------------------------------------------------------------ scanner
def test_base_folder_with_files_and_folders():
    # scan_files.rs:223-308 (paths need not exist: only the component walk is under test)
    assert probe("base", "/base/folder", "/base/folder/subfolder/file.jpg") == "/base"          # first arg seeds: parent of /base/folder
    cases = [(["/base/folder/x"], "/base/folder/subfolder/file.jpg", "/base/folder"),
             (["/base/folder/subfolder/another/folder/x"], "/base/folder/subfolder/file.jpg", "/base/folder/subfolder"),
             (["/base/folder/subfolder/another/folder/x"], "/file.jpg", "/"),
             (["/x"], "/base/folder/subfolder/file.jpg", "/"),
             (["/x"], "/file.jpg", "/")]
    for seed, new, want in cases:
        assert probe("base", *seed, new) == want, (seed, new)
    assert probe("base", "/temp/file.jpg") == "/temp"
    assert probe("base", "/") == "NONE"


def test_scan_and_filetype_detection(tree):
    root, files = tree
    # magic bytes: the liar and the text file are dropped; dry-run JSON lists what the scanner kept
    r = run_cli(EMUL_CLI, "-q", 80, "-o", root.parent / "o", "--json", "-d", "-R", root)
    got = [os.path.relpath(f["original_path"], root) for f in json.loads(r.stdout)["files"]]
    assert sorted(got) == sorted(files)
    # extension only: the liar is kept (and then fails in the engine, not in the scanner)
    r = run_cli(EMUL_CLI, "-q", 80, "-o", root.parent / "o", "--json", "-d", "-R", "--check-extension-only", root)
    got = [os.path.relpath(f["original_path"], root) for f in json.loads(r.stdout)["files"]]
    assert sorted(got) == sorted(list(files) + ["fake.jpg"])
    # non-recursive: only the top level
    r = run_cli(EMUL_CLI, "-q", 80, "-o", root.parent / "o", "--json", "-d", root)
    assert [os.path.basename(f["original_path"]) for f in json.loads(r.stdout)["files"]] == ["j0.JPG"]
    # explicit files + missing ones
    r = run_cli(EMUL_CLI, "-q", 80, "-o", root.parent / "o", "--json", "-d", root / "j0.JPG", root / "nope.jpg", root / "notes.txt")
    assert len(json.loads(r.stdout)["files"]) == 1


def test_no_files_and_no_base(tmp_path):
    r = run_cli(EMUL_CLI, "-q", 80, "-o", tmp_path)
    assert r.returncode == 0 and r.stderr.strip() == "No files to compress"
    r = run_cli(EMUL_CLI, "-q", 80, "-o", tmp_path, "--json")
    j = json.loads(r.stdout)
    assert j["error"] == "No files to compress" and j["files"] == [] and j["summary"]["total_files"] == 0
    r = run_cli(EMUL_CLI, "-q", 80, "-o", tmp_path, "--json", tmp_path / "missing.jpg")
    assert r.returncode == 255 and json.loads(r.stdout)["error"] == "Unable to compute the base path for the files."
    r = run_cli(EMUL_CLI, "-o", tmp_path, "x.jpg")
    assert r.returncode == 2 and "required" in r.stderr


# ------------------------------------------------------------------------------------------------ output paths
def test_output_path(tmp_path):
    # compressor.rs:615-766
    out, base = tmp_path / "output", tmp_path / "base"
    folder = base / "folder"
    out.mkdir(); folder.mkdir(parents=True)
    f = folder / "test.jpg"
    assert probe("outpath", out, f, base, 1, "_suffix", "original", 0).split("\n") == [str(out / "folder"), "test_suffix.jpg"]
    assert probe("outpath", out, f, base, 0, "_suffix", "original", 0).split("\n") == [str(out), "test_suffix.jpg"]
    assert probe("outpath", out, folder / "test", base, 0, "_suffix", "original", 0).split("\n") == [str(out), "test_suffix"]
    other = tmp_path / "different_base" / "folder"
    other.mkdir(parents=True)
    for fmt, ext in (("original", "jpg"), ("jpeg", "jpg"), ("png", "png"), ("webp", "webp"), ("tiff", "tiff"), ("gif", "gif")):
        assert probe("outpath", out, other / "test.jpg", base, 0, "_suffix", fmt, 0).split("\n") == [str(out), f"test_suffix.{ext}"]
    sub = folder / "subfolder"
    sub.mkdir()
    assert probe("outpath", out, sub / "test.jpg", base, 1, "_suffix", "original", 1).split("\n") == [str(sub), "test_suffix.jpg"]
    # keep_structure: a parent outside the base, or one that does not exist, has no output path
    assert probe("outpath", out, other / "test.jpg", base, 1, "", "original", 0) == "NONE"
    assert probe("outpath", out, tmp_path / "ghost" / "test.jpg", base, 1, "", "original", 0) == "NONE"


# ------------------------------------------------------------------------------------------------ recap + JSON
R3 = ["a.jpg", "out/a.jpg", 1000, 800, 0, "", "b.jpg", "out/b.jpg", 2000, 2000, 1, "skipped msg", "c.jpg", "", 500, 0, 2, "Error compressing file: x"]


def test_json_schema_and_statistics():
    # main.rs:643-727
    j = json.loads(probe("json", 0, *R3))
    assert list(j) == ["version", "dry_run", "error", "files", "summary"]
    assert j["version"] == "1.0.0" and j["dry_run"] is False and j["error"] is None
    assert [f["status"] for f in j["files"]] == ["success", "skipped", "error"]
    assert list(j["files"][0]) == ["original_path", "output_path", "original_size", "compressed_size", "status", "message"]
    s = j["summary"]
    assert list(s) == ["total_files", "success", "skipped", "errors", "original_size", "compressed_size", "savings_bytes", "savings_percent"]
    assert (s["total_files"], s["success"], s["skipped"], s["errors"]) == (3, 1, 1, 1)
    assert (s["original_size"], s["compressed_size"], s["savings_bytes"]) == (3500, 2800, 700) and abs(s["savings_percent"] - 20.0) < 1e-12
    raw = probe("json", 1)
    assert raw == '{"version":"1.0.0","dry_run":true,"error":null,"files":[],"summary":{"total_files":0,"success":0,"skipped":0,"errors":0,' \
                  '"original_size":0,"compressed_size":0,"savings_bytes":0,"savings_percent":0.0}}'
    # size increase -> negative savings; strings are escaped
    j = json.loads(probe("json", 0, 'we"ird\\name\n.jpg', "o", 100, 150, 0, ""))
    assert j["summary"]["savings_bytes"] == -50 and j["summary"]["savings_percent"] == -50.0 and j["files"][0]["original_path"] == 'we"ird\\name\n.jpg'


def test_recap_messages():
    # main.rs:433-541 + the ByteSize display format (bytesize 2.x: IEC units, one decimal)
    assert probe("recap", 3) == ""
    assert probe("recap", 0, *R3) == ""
    assert probe("recap", 1, *R3) == "Compressed 3 files (1 success, 1 skipped, 1 errors)\n3.4 KiB -> 2.7 KiB [-700 B | -20.00%]"
    v2 = probe("recap", 2, *R3)
    assert "[Success]" not in v2 and "[Skipped] b.jpg -> out/b.jpg\n2.0 KiB -> 2.0 KiB [-0 B | -0.00%]\nskipped msg\n" in v2
    assert "[Error] c.jpg -> \n500 B -> 0 B [-500 B | -100.00%]\nError compressing file: x\n" in v2
    v3 = probe("recap", 3, *R3)
    assert v3.startswith("[Success] a.jpg -> out/a.jpg\n1000 B -> 800 B [-200 B | -20.00%]\n\n[Skipped]")
    assert probe("recap", 1, "a", "b", 0, 0, 0, "") == "Compressed 1 files (1 success, 0 skipped, 0 errors)\n0 B -> 0 B [-0 B | -0.00%]"   # zero division
    assert probe("recap", 1, "a", "b", 100, 150, 0, "").endswith("[+50 B | +50.00%]")
    for n, text in ((1023, "1023 B"), (1024, "1.0 KiB"), (300_950, "293.9 KiB"), (1_048_576, "1.0 MiB"), (5 * 1024 ** 3, "5.0 GiB")):
        assert probe("fmtsize", n) == text


# ------------------------------------------------------------------------------------------------ dimensions
def test_probe_dimensions_and_exif_orientation(tmp_path):
    src = synth_jpeg(43, 120, 80)
    (tmp_path / "a.jpg").write_bytes(src)
    assert probe("dims", tmp_path / "a.jpg", 0) == "120 80"
    tiff = b"II*\x00\x08\x00\x00\x00" + b"\x01\x00" + b"\x12\x01\x03\x00\x01\x00\x00\x00\x06\x00\x00\x00" + b"\x00\x00\x00\x00"
    app1 = b"\xff\xe1" + (len(tiff) + 8).to_bytes(2, "big") + b"Exif\x00\x00" + tiff
    (tmp_path / "rot.jpg").write_bytes(src[:2] + app1 + src[2:])
    assert probe("dims", tmp_path / "rot.jpg", 1) == "80 120"      # orientation 6 swaps, but only with --exif
    assert probe("dims", tmp_path / "rot.jpg", 0) == "120 80"
    (tmp_path / "p.png").write_bytes(b"\x89PNG\r\n\x1a\n\x00\x00\x00\rIHDR" + (300).to_bytes(4, "big") + (200).to_bytes(4, "big") + b"\x08\x02\x00\x00\x00")
    assert probe("dims", tmp_path / "p.png", 0) == "300 200"
    (tmp_path / "g.gif").write_bytes(b"GIF89a" + (64).to_bytes(2, "little") + (48).to_bytes(2, "little") + b"\x00" * 8)
    assert probe("dims", tmp_path / "g.gif", 0) == "64 48"
    (tmp_path / "junk").write_bytes(b"\x00" * 64)
    assert probe("dims", tmp_path / "junk", 0) == "ERR"


# ------------------------------------------------------------------------------------------------ whole program
def end_to_end(binary, tree, tmp_path):
    root, files = tree
    out = tmp_path / "out"
    # 1. quality 80, keep structure, JSON: bytes equal the oracle's, results in input order
    r = run_cli(binary, "-q", 80, "-o", out, "-R", "-S", "--json", root)
    assert r.returncode == 0, r.stderr
    j = json.loads(r.stdout)
    assert [f["status"] for f in j["files"]] == ["success"] * 3
    for f in j["files"]:
        rel = os.path.relpath(f["original_path"], root)
        assert f["output_path"] == str(out / rel)
        data = open(f["output_path"], "rb").read()
        assert data == oracle_lossy(files[rel], 80) and f["compressed_size"] == len(data) and f["original_size"] == len(files[rel])
    # 2. overwrite never / bigger: everything is skipped, sizes report the original
    for policy in ("never", "bigger"):
        j = json.loads(run_cli(binary, "-q", 80, "-o", out, "-R", "-S", "--json", "-O", policy, root).stdout)
        assert [f["status"] for f in j["files"]] == ["skipped"] * 3
        assert all(f["message"] == "File already exists, skipped due overwrite policy" and f["compressed_size"] == f["original_size"] for f in j["files"])
    # 3. dry run writes nothing
    dry = tmp_path / "dry"
    j = json.loads(run_cli(binary, "-q", 80, "-o", dry, "-R", "--json", "-d", root).stdout)
    assert j["dry_run"] is True and not dry.exists() and all(f["status"] == "success" and f["compressed_size"] == f["original_size"] for f in j["files"])
    # 4. lossless + suffix + flat output + keep dates
    old = time.time() - 86400 * 30
    for rel in files:
        os.utime(root / rel, (old, old))
    flat = tmp_path / "flat"
    j = json.loads(run_cli(binary, "--lossless", "-o", flat, "-R", "--json", "--suffix", "_c", "--keep-dates", root).stdout)
    for f in j["files"]:
        rel = os.path.relpath(f["original_path"], root)
        stem, ext = os.path.splitext(os.path.basename(rel))
        assert f["output_path"] == str(flat / f"{stem}_c{ext}")
        assert open(f["output_path"], "rb").read() == oracle_lossless(files[rel])
        assert abs(os.stat(f["output_path"]).st_mtime - old) < 2
    # 5. --max-size: under the limit (or the smallest try); 6. min-savings skip
    lim = len(files["j0.JPG"]) // 3
    j = json.loads(run_cli(binary, "--max-size", lim, "-o", tmp_path / "ms", "--json", root / "j0.JPG").stdout)
    assert j["files"][0]["status"] == "success" and j["files"][0]["compressed_size"] <= lim
    j = json.loads(run_cli(binary, "-q", 80, "-o", tmp_path / "sv", "--json", "--min-savings", "99%", root / "j0.JPG").stdout)
    assert j["files"][0]["status"] == "skipped" and j["files"][0]["message"].startswith("Insufficient savings: ") and j["files"][0]["message"].endswith("% < 99.00%, skipped")
    assert not (tmp_path / "sv" / "j0.JPG").exists()
    j = json.loads(run_cli(binary, "-q", 80, "-o", tmp_path / "sv", "--json", "--min-savings", "1MB", root / "j0.JPG").stdout)
    assert j["files"][0]["message"].endswith(" < 976.6 KiB, skipped")
    # 7. long edge: landscape gets width, portrait gets height (compressor.rs:935-983), pixels equal the oracle's resize
    j = json.loads(run_cli(binary, "-q", 80, "-o", tmp_path / "le", "-R", "--json", "--long-edge", 100, root / "j0.JPG", root / "level_1_0" / "j1.jpg").stdout)
    assert open(j["files"][0]["output_path"], "rb").read() == oracle_resized(files["j0.JPG"], 100, 0)
    assert open(j["files"][1]["output_path"], "rb").read() == oracle_resized(files["level_1_0/j1.jpg"], 0, 100)
    # --no-upscale turns an enlarging request into a plain compression (compressor.rs:899-932)
    j = json.loads(run_cli(binary, "-q", 80, "-o", tmp_path / "nu", "--json", "--width", 4000, "--no-upscale", root / "j0.JPG").stdout)
    assert open(j["files"][0]["output_path"], "rb").read() == oracle_lossy(files["j0.JPG"], 80)
    # 8. engine errors are per file; same-folder output; the recap on stdout
    r = run_cli(binary, "-q", 80, "--same-folder-as-input", "--suffix", ".min", "--check-extension-only", "--verbose", 2, root)
    assert "[Error] " in r.stdout and "fake.jpg" in r.stdout and "Error compressing file: " in r.stdout
    assert r.stdout.rstrip().split("\n")[-2] == "Compressed 2 files (1 success, 0 skipped, 1 errors)"
    assert (root / "j0.min.JPG").read_bytes() == oracle_lossy(files["j0.JPG"], 80)
    # 9. formats without a device path fail per file, not the run
    j = json.loads(run_cli(binary, "-q", 80, "-o", tmp_path / "cv", "--json", "--format", "tiff", root / "level_1_0" / "j1.jpg").stdout)
    assert j["files"][0]["status"] == "error" and j["files"][0]["message"].startswith("Error compressing file: ") and j["files"][0]["output_path"].endswith("j1.tiff")


    # 9b. JPEG -> WebP is built (configs[3] shape: convert + long edge): bytes equal the oracle's
    from _util import oracle_jpeg_to_webp
    j = json.loads(run_cli(binary, "-q", 85, "-o", tmp_path / "wp", "--json", "--format", "webp", "--long-edge", 90, root / "level_1_0" / "j1.jpg", root / "j0.JPG").stdout)
    assert [f["status"] for f in j["files"]] == ["success"] * 2 and j["files"][0]["output_path"].endswith("j1.webp")
    assert open(j["files"][0]["output_path"], "rb").read() == oracle_jpeg_to_webp(files["level_1_0/j1.jpg"], 85, 0, 90)
    assert open(j["files"][1]["output_path"], "rb").read() == oracle_jpeg_to_webp(files["j0.JPG"], 85, 90, 0)
    # 10. PNG: --lossless runs the device PNG pipeline (mixed with JPEG in one run, order kept)
    from _util import oracle_png
    from gen_synth import synth_png
    mixed = tmp_path / "mixed"
    mixed.mkdir()
    pngs = {"a.png": synth_png(60, 120, 80, "RGB", compress_level=1), "c.png": synth_png(61, 64, 64, "LA", compress_level=1)}
    for name, data in pngs.items():
        (mixed / name).write_bytes(data)
    (mixed / "b.jpg").write_bytes(files["level_1_0/j1.jpg"])
    j = json.loads(run_cli(binary, "--lossless", "-o", tmp_path / "mx", "--json", "--png-opt-level", 2, mixed).stdout)
    assert [os.path.basename(f["original_path"]) for f in j["files"]] == ["a.png", "b.jpg", "c.png"]
    assert [f["status"] for f in j["files"]] == ["success"] * 3
    for f in j["files"]:
        name = os.path.basename(f["original_path"])
        want = oracle_png(pngs[name], 2) if name in pngs else oracle_lossless(files["level_1_0/j1.jpg"])
        assert open(f["output_path"], "rb").read() == want, name


def lossy_png_step(binary, tmp_path):
    """-q on a PNG: the lossy (quantising) form of the PNG pipeline.  Its own step: on the device it runs from tests/test_zz_png_lossy_gpu.py"""
    from _util import oracle_png_lossy
    from gen_synth import synth_png
    d = tmp_path / "lossy_in"
    d.mkdir()
    png = synth_png(60, 120, 80, "RGB", compress_level=1)
    (d / "a.png").write_bytes(png)
    j = json.loads(run_cli(binary, "-q", 80, "-o", tmp_path / "mq", "--json", d / "a.png").stdout)
    assert j["files"][0]["status"] == "success" and open(j["files"][0]["output_path"], "rb").read() == oracle_png_lossy(png)
    # --max-size on a PNG: the bisection over png.quality
    from test_pipeline_emul import reference_size_walk
    target = len(oracle_png_lossy(png, quality=30)) + 30
    j = json.loads(run_cli(binary, "--max-size", target, "-o", tmp_path / "mq2", "--json", d / "a.png").stdout)
    want = reference_size_walk(png, target, encode=lambda s, q: oracle_png_lossy(s, quality=q))[1]
    assert j["files"][0]["status"] == "success" and open(j["files"][0]["output_path"], "rb").read() == want and len(want) <= target


def png_to_webp_step(binary, tmp_path):
    """--format webp over PNG and JPEG sources in one run.  Its own step: on the device it runs from tests/test_zzz_png_webp_gpu.py"""
    from _util import oracle_jpeg_to_webp
    from gen_synth import synth_jpeg, synth_png
    from oracle import oracle as O
    d = tmp_path / "pw_in"
    d.mkdir()
    png, grey, rgba, jpg = synth_png(61, 100, 70, "RGB"), synth_png(62, 33, 50, "L"), synth_png(63, 40, 30, "RGBA"), synth_jpeg(64, 96, 64, texture=5)
    for name, data in (("a.png", png), ("b.png", grey), ("c.png", rgba), ("d.jpg", jpg)):
        (d / name).write_bytes(data)
    j = json.loads(run_cli(binary, "-q", 70, "-o", tmp_path / "pw", "--json", "--format", "webp", d / "a.png", d / "b.png", d / "c.png", d / "d.jpg").stdout)
    assert [f["status"] for f in j["files"]] == ["success", "success", "success", "success"]
    got = [open(f["output_path"], "rb").read() for f in j["files"]]
    assert [got[0], got[1], got[3]] == [O.png_to_webp(png, 70), O.png_to_webp(grey, 70), oracle_jpeg_to_webp(jpg, 70)]
    assert got[2][12:16] == b"VP8X" and b"ALPH" in got[2][:64]   # the RGBA picture keeps its alpha (tests/test_png_webp_emul.py checks the chunk)
    assert j["files"][0]["output_path"].endswith("a.webp")
    from _util import oracle_png_to_webp
    j = json.loads(run_cli(binary, "-q", 70, "-o", tmp_path / "pw2", "--json", "--format", "webp", "--long-edge", 60, d / "a.png", d / "d.jpg").stdout)
    assert [f["status"] for f in j["files"]] == ["success", "success"]
    assert [open(f["output_path"], "rb").read() for f in j["files"]] == [oracle_png_to_webp(png, 70, 60, 0), oracle_jpeg_to_webp(jpg, 70, 60, 0)]
    # --format webp --max-size, as the reference does it (compressor.rs:287-296): convert at the default quality, then the size walk over the
    # WebP that came out -- every try decodes it (the VP8 decoder; libwebp through Pillow states what it must give) and encodes the pixels again
    import io

    import numpy as np
    from PIL import Image
    from oracle import oracle as O
    from test_pipeline_emul import reference_size_walk
    target = len(oracle_jpeg_to_webp(jpg, 40)) + 25
    j = json.loads(run_cli(binary, "--max-size", target, "-o", tmp_path / "pw3", "--json", "--format", "webp", d / "d.jpg", d / "a.png").stdout)
    assert [f["status"] for f in j["files"]] == ["success", "success"]
    want = []
    for first in (oracle_jpeg_to_webp(jpg, 80), oracle_png_to_webp(png, 80)):
        rgb = np.ascontiguousarray(np.asarray(Image.open(io.BytesIO(first)).convert("RGB")))
        want.append(reference_size_walk(first, target, encode=lambda s_, q: O.webp_encode_rgb(rgb, q))[1])
    assert [open(f["output_path"], "rb").read() for f in j["files"]] == want and len(want[0]) <= target


def jpeg_to_png_step(binary, tmp_path):
    """--format png over JPEG sources: lossless target, quantising target, with a resize.  Its own step: on the device it runs from
    tests/test_zzz_jpeg_png_gpu.py"""
    from _util import oracle_jpeg_to_png
    from gen_synth import synth_jpeg
    d = tmp_path / "jp_in"
    d.mkdir()
    a, b = synth_jpeg(71, 120, 80, texture=6), synth_jpeg(72, 64, 96, subsampling=0, texture=3)
    (d / "a.jpg").write_bytes(a); (d / "b.jpg").write_bytes(b)
    j = json.loads(run_cli(binary, "--lossless", "-o", tmp_path / "jp1", "--json", "--format", "png", d / "a.jpg", d / "b.jpg").stdout)
    assert [f["status"] for f in j["files"]] == ["success"] * 2 and j["files"][0]["output_path"].endswith("a.png")
    assert [open(f["output_path"], "rb").read() for f in j["files"]] == [oracle_jpeg_to_png(a, True), oracle_jpeg_to_png(b, True)]
    j = json.loads(run_cli(binary, "-q", 80, "-o", tmp_path / "jp2", "--json", "--format", "png", "--long-edge", 60, d / "a.jpg", d / "b.jpg").stdout)
    assert [f["status"] for f in j["files"]] == ["success"] * 2
    assert [open(f["output_path"], "rb").read() for f in j["files"]] == [oracle_jpeg_to_png(a, False, 3, 60, 0), oracle_jpeg_to_png(b, False, 3, 0, 60)]


def png_resize_step(binary, tmp_path):
    """--long-edge over PNG and JPEG sources in one run (one parameter group per orientation).  Its own step: on the device it runs from
    tests/test_zzz_png_resize_gpu.py"""
    from _util import oracle_png_resized, oracle_resized
    from gen_synth import synth_jpeg, synth_png
    d = tmp_path / "pr_in"
    d.mkdir()
    wide, tall, pal, jpg = synth_png(81, 120, 70, "RGB"), synth_png(82, 50, 90, "RGBA"), synth_png(83, 60, 40, "P"), synth_jpeg(84, 100, 60, texture=4)
    for name, data in (("a.png", wide), ("b.png", tall), ("c.png", pal), ("d.jpg", jpg)):
        (d / name).write_bytes(data)
    j = json.loads(run_cli(binary, "--lossless", "--png-opt-level", 2, "-o", tmp_path / "pr", "--json", "--long-edge", 45, d / "a.png", d / "b.png", d / "c.png").stdout)
    assert [f["status"] for f in j["files"]] == ["success", "success", "success"]
    got = [open(f["output_path"], "rb").read() for f in j["files"]]
    assert got == [oracle_png_resized(wide, True, 2, 45, 0), oracle_png_resized(tall, True, 2, 0, 45), oracle_png_resized(pal, True, 2, 45, 0)]
    j = json.loads(run_cli(binary, "-q", 80, "-o", tmp_path / "pr2", "--json", "--width", 40, d / "a.png", d / "d.jpg").stdout)
    assert [f["status"] for f in j["files"]] == ["success", "success"]
    assert [open(f["output_path"], "rb").read() for f in j["files"]] == [oracle_png_resized(wide, False, 3, 40, 0), oracle_resized(jpg, 40, 0)]


def png_to_jpeg_step(binary, tmp_path):
    """--format jpeg over PNG sources, with and without a resize.  Its own step: on the device it runs from tests/test_zzzz_png_jpeg_gpu.py"""
    from _util import oracle_png_to_jpeg
    from gen_synth import synth_png
    d = tmp_path / "pj_in"
    d.mkdir()
    rgb, rgba, pal = synth_png(91, 120, 70, "RGB"), synth_png(92, 50, 90, "RGBA"), synth_png(93, 60, 40, "P")
    for name, data in (("a.png", rgb), ("b.png", rgba), ("c.png", pal)):
        (d / name).write_bytes(data)
    j = json.loads(run_cli(binary, "-q", 70, "-o", tmp_path / "pj", "--json", "--format", "jpeg", d / "a.png", d / "b.png", d / "c.png").stdout)
    assert [f["status"] for f in j["files"]] == ["success"] * 3 and j["files"][0]["output_path"].endswith("a.jpg")
    assert [open(f["output_path"], "rb").read() for f in j["files"]] == [oracle_png_to_jpeg(s, 70) for s in (rgb, rgba, pal)]
    j = json.loads(run_cli(binary, "-q", 85, "-o", tmp_path / "pj2", "--json", "--format", "jpeg", "--long-edge", 48, "--jpeg-baseline", "--jpeg-chroma-subsampling", "4:4:4",
                           d / "a.png", d / "b.png").stdout)
    assert [f["status"] for f in j["files"]] == ["success"] * 2
    want = [oracle_png_to_jpeg(rgb, 85, 48, 0, 444, 0), oracle_png_to_jpeg(rgba, 85, 0, 48, 444, 0)]
    assert [open(f["output_path"], "rb").read() for f in j["files"]] == want


def test_tree_in_windows_equals_tree_in_one_piece(tree, tmp_path):
    """the CLI takes a tree a window of files at a time (bounded memory); CSH_CLI_WINDOW=1 and =2 must give the files, the JSON and the order of the default run"""
    root, _ = tree
    runs = []
    for label, window in (("whole", None), ("w1", "1"), ("w2", "2")):
        out = tmp_path / ("out_" + label)
        env = dict(os.environ)
        if window:
            env["CSH_CLI_WINDOW"] = window
        r = subprocess.run([EMUL_CLI, "-q", "80", "-R", "-S", "--json", "-o", str(out), str(root)], capture_output=True, text=True, env=env)
        assert r.returncode == 0, r.stderr
        j = json.loads(r.stdout)
        listing = {os.path.relpath(os.path.join(d, f), out): open(os.path.join(d, f), "rb").read() for d, _, fs in os.walk(out) for f in fs}
        runs.append(([(os.path.relpath(f["original_path"], root), f["status"], f["compressed_size"]) for f in j["files"]], listing))
    assert runs[0] == runs[1] == runs[2] and len(runs[0][0]) >= 3 and len(runs[0][1]) >= 3


def test_read_ahead_stops_when_an_output_is_a_later_input(tmp_path):
    """the window read-ahead must not read a later window's input while an earlier window is still writing that very file
    (--suffix / --format into the input tree).  a.jpg -> a_z.jpg, and a_z.jpg is itself an input of the next window: the sequential reading is
    a_z_z.jpg = compress(compress(a.jpg)); a reader running ahead would have compressed the old a_z.jpg."""
    d = tmp_path / "ra"
    d.mkdir()
    a, stale = synth_jpeg(61, 96, 64, texture=30), synth_jpeg(62, 80, 48, texture=5)
    (d / "a.jpg").write_bytes(a)
    (d / "a_z.jpg").write_bytes(stale)
    env = dict(os.environ, CSH_CLI_WINDOW="1")
    r = subprocess.run([EMUL_CLI, "-q", "80", "--suffix", "_z", "--same-folder-as-input", "-O", "all", "--json", str(d / "a.jpg"), str(d / "a_z.jpg")], capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr
    j = json.loads(r.stdout)
    assert [os.path.basename(f["output_path"]) for f in j["files"]] == ["a_z.jpg", "a_z_z.jpg"]
    once = oracle_lossy(a)
    assert (d / "a_z.jpg").read_bytes() == once and (d / "a_z_z.jpg").read_bytes() == oracle_lossy(once)


def gpus_dealing(binary, tmp_path):
    """--gpus 3 on a box with one device (CSH_CLI_SAME_DEVICE=1: three device slots, all device 0): the batches are dealt over 3 x CSH_CLI_WORKERS host threads,
    small batches so that every slot gets several; files, JSON and order are those of the plain run"""
    d = tmp_path / "gp_in"
    d.mkdir()
    srcs = [synth_jpeg(300 + i, 96 + 8 * (i % 3), 64, texture=10 + i) for i in range(23)]
    for i, s in enumerate(srcs):
        (d / f"g{i:02d}.jpg").write_bytes(s)
    runs = []
    for label, extra, env in (("one", [], {}), ("three", ["--gpus", "3"], {"CSH_CLI_SAME_DEVICE": "1", "CSH_CLI_BATCH": "2"})):
        out = tmp_path / ("gp_" + label)
        r = subprocess.run([binary, "-q", "75", "--json", "-o", str(out), *extra, str(d)], capture_output=True, text=True, env=dict(os.environ, **env))
        assert r.returncode == 0, r.stderr
        j = json.loads(r.stdout)
        runs.append([(os.path.basename(f["original_path"]), f["status"], open(f["output_path"], "rb").read()) for f in j["files"]])
    assert runs[0] == runs[1] and len(runs[0]) == 23
    assert [x[2] for x in runs[0]] == [oracle_lossy(s, 75) for s in srcs]


def test_gpus_flag_deals_batches_over_device_slots(tmp_path):
    gpus_dealing(EMUL_CLI, tmp_path)


def test_whole_program_emulated(tree, tmp_path):
    end_to_end(EMUL_CLI, tree, tmp_path)
    lossy_png_step(EMUL_CLI, tmp_path)
    png_to_webp_step(EMUL_CLI, tmp_path)
    jpeg_to_png_step(EMUL_CLI, tmp_path)
    png_resize_step(EMUL_CLI, tmp_path)
    png_to_jpeg_step(EMUL_CLI, tmp_path)


@pytest.mark.gpu
def test_whole_program_on_device(tree, tmp_path):
    assert os.path.exists(PRODUCT_CLI), "caesium-clt_amd/bin/caesiumclt is not built (python -c 'import __graft_entry__ as g; g.build()')"
    end_to_end(PRODUCT_CLI, tree, tmp_path)
    gpus_dealing(PRODUCT_CLI, tmp_path)
    # many small files across two parameter groups in one run, order preserved
    many = tmp_path / "many"
    many.mkdir()
    srcs = []
    for i in range(24):
        w, h = (160, 96) if i % 2 else (96, 160)
        srcs.append(synth_jpeg(100 + i, w, h, texture=20))
        (many / f"f{i:02d}.jpg").write_bytes(srcs[-1])
    j = json.loads(run_cli(PRODUCT_CLI, "-q", 70, "-o", tmp_path / "mo", "--json", "--long-edge", 80, many).stdout)
    assert [os.path.basename(f["original_path"]) for f in j["files"]] == [f"f{i:02d}.jpg" for i in range(24)]
    for i, f in enumerate(j["files"]):
        want = oracle_resized(srcs[i], 0 if i % 2 == 0 else 80, 80 if i % 2 == 0 else 0, quality=70)
        assert open(f["output_path"], "rb").read() == want, i


def test_product_binary_fails_loudly_without_gpu(tree, tmp_path):
    """the product CLI has no CPU path: on a box without a device every file is an error naming the missing device"""
    import torch
    if torch.cuda.is_available() or not os.path.exists(PRODUCT_CLI):
        pytest.skip("needs the product binary on a GPU-less box")
    root, _ = tree
    j = json.loads(run_cli(PRODUCT_CLI, "-q", 80, "-o", tmp_path / "o", "--json", root).stdout)
    assert j["files"] and all(f["status"] == "error" and "device" in f["message"].lower() for f in j["files"])


def test_transparency_through_the_cli(tmp_path):
    """WebP files with transparency (an ALPH chunk / a VP8L picture that is not opaque) and a transparent PNG through the whole program on the emulation build:
    -q with a resize keeps the alpha (the VP8X canvas size is what the resize parameters are computed from), --format webp on the PNG as well"""
    import io

    import numpy as np
    from PIL import Image

    from gen_synth import synth_rgb
    d = tmp_path / "alpha_in"
    d.mkdir()
    rgb = synth_rgb(1, 80, 56, texture=10.0)
    a = np.tile((np.arange(80) * 3).astype(np.uint8), (56, 1))
    for name, kw in (("a.webp", dict(quality=80)), ("b.webp", dict(lossless=True))):
        Image.fromarray(np.dstack([rgb, a]), "RGBA").save(d / name, format="WEBP", **kw)
    Image.fromarray(np.dstack([rgb, a]), "RGBA").save(d / "c.png")
    j = json.loads(run_cli(EMUL_CLI, "-q", 70, "--long-edge", 40, "-o", tmp_path / "alpha_out", "--json", d / "a.webp", d / "b.webp", d / "c.png").stdout)
    assert [f["status"] for f in j["files"]] == ["success"] * 3
    for f in j["files"][:2]:
        im = Image.open(f["output_path"])
        assert (im.format, im.mode, im.size) == ("WEBP", "RGBA", (40, 28))
    j = json.loads(run_cli(EMUL_CLI, "-q", 70, "--format", "webp", "-o", tmp_path / "alpha_out2", "--json", d / "c.png").stdout)
    assert j["files"][0]["status"] == "success"
    im = Image.open(j["files"][0]["output_path"])
    assert (im.format, im.mode, im.size) == ("WEBP", "RGBA", (80, 56)) and np.array_equal(np.asarray(im)[:, :, 3], a)
    j = json.loads(run_cli(EMUL_CLI, "--lossless", "-o", tmp_path / "alpha_out3", "--json", d / "b.webp").stdout)
    assert j["files"][0]["status"] == "success"
    assert np.array_equal(np.asarray(Image.open(j["files"][0]["output_path"])), np.asarray(Image.open(d / "b.webp")))   # (libwebp itself cleared the colour under alpha 0 when it made b.webp)
