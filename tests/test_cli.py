"""D1/D2: the console driver (flags, help, pairs.txt handling, PNG codec). CPU tests cover the parts that need no GPU;
the GPU test runs a real pairs.txt batch and compares the written PNGs with nct_process_pair's result."""
import os
import subprocess
import numpy as np
import pytest
from PIL import Image

import nct
import synth

BIN = os.path.join(nct.PKG_ROOT, "bin", "neural_color_transfer")
REF_DEMO = "/root/reference/demo/example/in"


def run(*args):
    return subprocess.run([BIN, *args], capture_output=True, text=True)


def test_help_prints_flags_and_returns_minus_one():
    for h in ("-h", "-?", "-help", "/h"):
        r = run(h)
        assert r.returncode == 255                               # main.cu:557-560: `return -1`
        for flag in ("-m:", "-i:", "-o:", "-g:", "-bds:", "-eps:", "-nl:", "-l:", "-w:"):
            assert flag in r.stdout
    # CmdLine.h:140-142 prints "-<flag>: (default=<value that applies>) <comment>"; the value is Config::Config()'s (Config.h:58-72),
    # the comment strings are the reference's own, stale remarks included (main.cu:40-43)
    assert "-l: (default=0.125) Weight of local constraitn (default: 0.001)." in r.stdout
    assert "-w: (default=0.024) " in r.stdout and "-bds: (default=2) " in r.stdout and "-g: (default=0) GPU ID" in r.stdout
    assert "-m: (default=) Directory of network models." in r.stdout


def test_unknown_flag():
    r = run("-zzz", "1")
    assert r.returncode == 255 and "Unrecognized parameter: -zzz" in r.stdout and "-bds:" in r.stdout


def test_missing_pairs_txt_is_an_error_not_a_crash(tmp_path):
    r = run("-i", str(tmp_path), "-o", str(tmp_path / "out"), "-m", str(tmp_path))
    assert r.returncode == 255 and "pairs.txt does not exist" in r.stdout


@pytest.mark.parametrize("mode", ["RGB", "RGBA", "L", "P", "LA", "I;16"])
def test_png_codec_roundtrip_matches_pillow(tmp_path, mode):
    """imread semantics: 8-bit 3-channel BGR, alpha dropped (not blended), gray replicated, palette expanded, 16->8 bit."""
    rgb = synth.image(3, 37, 53)[..., ::-1].copy()
    if mode == "RGB":
        im = Image.fromarray(rgb, "RGB"); exp = rgb
    elif mode == "RGBA":
        a = np.full(rgb.shape[:2] + (1,), 77, np.uint8); im = Image.fromarray(np.concatenate([rgb, a], -1), "RGBA"); exp = rgb
    elif mode == "L":
        g = rgb[..., 0]; im = Image.fromarray(g, "L"); exp = np.stack([g] * 3, -1)
    elif mode == "LA":
        g = rgb[..., 1]; im = Image.fromarray(np.stack([g, 255 - g], -1), "LA"); exp = np.stack([g] * 3, -1)
    elif mode == "P":
        im = Image.fromarray(rgb, "RGB").quantize(64); exp = np.asarray(im.convert("RGB"))
    else:
        g16 = (rgb[..., 0].astype(np.uint16) << 8) | rgb[..., 1]; im = Image.fromarray(g16, "I;16"); exp = np.stack([rgb[..., 0]] * 3, -1)
    src, dst = str(tmp_path / "in.png"), str(tmp_path / "out.png")
    im.save(src)
    r = run("--png-roundtrip", src, dst)
    assert r.returncode == 0, r.stdout
    assert r.stdout.split() == ["53", "37"]
    got = np.asarray(Image.open(dst).convert("RGB"))
    assert np.array_equal(got, exp)


@pytest.mark.skipif(not os.path.isdir(REF_DEMO), reason="reference demo images not mounted")
def test_png_codec_reads_the_reference_demo_inputs(tmp_path):
    """demo/example/in/*.png (5 of 10 are RGBA) decode to the same pixels Pillow sees."""
    for name in sorted(os.listdir(REF_DEMO))[:10]:
        if not name.endswith(".png"):
            continue
        dst = str(tmp_path / name)
        r = run("--png-roundtrip", os.path.join(REF_DEMO, name), dst)
        assert r.returncode == 0, (name, r.stdout)
        assert np.array_equal(np.asarray(Image.open(dst).convert("RGB")), np.asarray(Image.open(os.path.join(REF_DEMO, name)).convert("RGB")))


@pytest.mark.parametrize("sub,q", [(0, 92), (1, 85), (2, 75), ("gray", 90), (2, 40)])
def test_jpeg_decoder_matches_libjpeg(tmp_path, sub, q):
    """cv::imread's JPEG path restated (host/jpeg_io.h): islow IDCT, fancy 2x1 / 2x2 chroma upsampling, fixed-point YCbCr->RGB —
    bit-identical to Pillow's libjpeg(-turbo) decode for 4:4:4 / 4:2:2 / 4:2:0 / grayscale, odd sizes and restart markers."""
    for (h, w) in ((64, 64), (61, 75), (17, 33), (123, 200)):
        rgb = synth.image(5 + h, h, w)[..., ::-1].copy()
        for kw in ({}, {"restart_marker_blocks": 3}):
            src = str(tmp_path / f"t{h}_{w}_{len(kw)}.jpg")
            im = Image.fromarray(rgb)
            try:
                if sub == "gray":
                    im.convert("L").save(src, quality=q, **kw)
                else:
                    im.save(src, quality=q, subsampling=sub, **kw)
            except TypeError:
                continue                                                   # older Pillow without restart_marker_blocks
            exp = np.asarray(Image.open(src).convert("RGB"))
            dst = str(tmp_path / "o.png")
            r = run("--png-roundtrip", src, dst)
            assert r.returncode == 0, r.stdout
            assert np.array_equal(np.asarray(Image.open(dst).convert("RGB")), exp), (h, w, sub, q, kw)


@pytest.mark.parametrize("sub,q", [(0, 90), (2, 75), (1, 30), ("gray", 85)])
def test_jpeg_progressive_matches_libjpeg(tmp_path, sub, q):
    """Progressive JPEG (SOF2: DC / AC first and refinement scans, EOB runs, restart intervals — jdphuff.c restated): bit-identical to Pillow."""
    for (h, w) in ((64, 64), (61, 75), (200, 150)):
        rgb = synth.image(9 + h, h, w)[..., ::-1].copy()
        for kw in ({}, {"restart_marker_blocks": 5}):
            src = str(tmp_path / f"p{h}_{w}_{len(kw)}.jpg")
            im = Image.fromarray(rgb)
            try:
                if sub == "gray":
                    im.convert("L").save(src, quality=q, progressive=True, **kw)
                else:
                    im.save(src, quality=q, subsampling=sub, progressive=True, **kw)
            except TypeError:
                continue
            exp = np.asarray(Image.open(src).convert("RGB"))
            dst = str(tmp_path / "o.png")
            r = run("--png-roundtrip", src, dst)
            assert r.returncode == 0, r.stdout
            assert np.array_equal(np.asarray(Image.open(dst).convert("RGB")), exp), (h, w, sub, q, kw)


def test_jpeg_truncated_file_is_an_error_or_decodes_without_crashing(tmp_path):
    src = tmp_path / "t.jpg"
    Image.fromarray(synth.image(1, 80, 80)).save(src, quality=80)
    raw = src.read_bytes()
    for cut in (10, 200, len(raw) // 2):
        (tmp_path / "c.jpg").write_bytes(raw[:cut])
        r = run("--png-roundtrip", str(tmp_path / "c.jpg"), str(tmp_path / "o.png"))
        assert r.returncode in (0, 1)                                      # never a signal


def test_png_reader_rejects_malformed_files(tmp_path):
    """Untrusted inputs: short / misplaced IHDR, corrupted chunk CRC, absurd dimensions — an error message, never a crash."""
    import struct, zlib
    good = tmp_path / "g.png"
    Image.fromarray(synth.image(1, 20, 24)).save(good)
    raw = good.read_bytes()

    def chunk(t, body):
        return struct.pack(">I", len(body)) + t + body + struct.pack(">I", zlib.crc32(t + body))
    cases = {
        "crc": raw[:40] + bytes([raw[40] ^ 0x55]) + raw[41:],                                   # flipped byte inside IDAT
        "short_ihdr": raw[:8] + chunk(b"IHDR", b"\0" * 5) + raw[33:],
        "late_ihdr": raw[:8] + chunk(b"tEXt", b"a\0b") + raw[8:],
        "huge": raw[:8] + chunk(b"IHDR", struct.pack(">IIBBBBB", 16000, 16000, 16, 6, 0, 0, 0)) + raw[33:],
    }
    for name, data in cases.items():
        p = tmp_path / f"{name}.png"
        p.write_bytes(data)
        r = run("--png-roundtrip", str(p), str(tmp_path / "o.png"))
        assert r.returncode == 1 and "Error" in r.stdout, name


def test_decoders_survive_mutated_files_under_sanitizers(tmp_path):
    """tests/fuzz_decoders.cpp: 20 000 mutations (bit flips, 0xFF runs, truncation, splices, header bytes) of PNG / JPEG seed files
    through host/png_io.h + host/jpeg_io.h built with AddressSanitizer + UBSan: every file decodes or is rejected with a message.
    (Found and fixed in round 2: DC Huffman categories > 15 and oversubscribed code lengths in DHT, negative left shifts in the IDCT.)"""
    a = synth.image(5, 48, 40)[..., ::-1].copy()
    seeds = []
    def put(name, im, **kw):
        im.save(tmp_path / name, **kw); seeds.append(str(tmp_path / name))
    put("a.png", Image.fromarray(a)); put("g.png", Image.fromarray(a).convert("L")); put("p.png", Image.fromarray(a).convert("P"))
    put("rgba.png", Image.fromarray(a).convert("RGBA")); put("i16.png", Image.fromarray((a[..., 0].astype(np.uint16) * 257)))
    put("a.jpg", Image.fromarray(a), quality=85); put("pr.jpg", Image.fromarray(a), quality=80, progressive=True)
    put("s.jpg", Image.fromarray(a), quality=90, subsampling=2); put("g.jpg", Image.fromarray(a).convert("L"), quality=75)
    exe = str(tmp_path / "fuzz")
    host = os.path.join(nct.PKG_ROOT, "host")
    r = subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-I", host,
                        os.path.join(os.path.dirname(__file__), "fuzz_decoders.cpp"), "-o", exe, "-lz"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([exe, "20000", "7", str(tmp_path / "scratch.bin")] + seeds, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    decoded, rejected = [int(x) for x in r.stdout.split()[1::2]]
    assert decoded + rejected == 20000 and decoded > 1000 and rejected > 1000


def test_gpu_locality_from_sysfs(tmp_path):
    """`-pin`: the CPUs next to a GPU come from /sys/bus/pci/devices/<addr>/{numa_node,local_cpulist} (host/affinity.h); checked against a fake sysfs tree
    through the CLI's self-test hook (NCT_SYSFS_ROOT), incl. the cpulist grammar and the node-cpulist fallback."""
    root = tmp_path / "sys"
    d = root / "bus" / "pci" / "devices" / "0000:c1:00.0"; d.mkdir(parents=True)
    (d / "numa_node").write_text("1\n"); (d / "local_cpulist").write_text("32-63,160-191\n")
    env = dict(os.environ, NCT_SYSFS_ROOT=str(root))
    r = subprocess.run([BIN, "--gpu-locality", "0000:c1:00.0"], capture_output=True, text=True, env=env)
    assert r.returncode == 0 and r.stdout.strip() == "1 32-63,160-191"
    d2 = root / "bus" / "pci" / "devices" / "0000:05:00.0"; d2.mkdir(parents=True)
    (d2 / "numa_node").write_text("0\n"); (d2 / "local_cpulist").write_text("\n")          # empty list -> falls back to the node's cpulist
    n0 = root / "devices" / "system" / "node" / "node0"; n0.mkdir(parents=True); (n0 / "cpulist").write_text("0-3,8\n")
    r = subprocess.run([BIN, "--gpu-locality", "0000:05:00.0"], capture_output=True, text=True, env=env)
    assert r.stdout.strip() == "0 0-3,8"
    r = subprocess.run([BIN, "--gpu-locality", "0000:ff:00.0"], capture_output=True, text=True, env=env)      # unknown device: no pinning, no failure
    assert r.returncode == 0 and r.stdout.strip() == "-1"
    (d / "local_cpulist").write_text("7-3\n")                                                   # malformed range -> ignored
    r = subprocess.run([BIN, "--gpu-locality", "0000:c1:00.0"], capture_output=True, text=True, env=env)
    assert r.stdout.strip() == "1"


def test_deploy_prototxt_check(tmp_path):
    """Classifier builds its Net from the deploy prototxt (Classifier.cpp:16); the library's topology is built in, so the file is checked: the VGG19 files pass
    (V2 `layer` + V1 `layers` spelling, comments, trailing fc layers ignored), anything that is another network up to relu5_1 is refused with a reason."""
    from caffemodel_io import write_deploy_prototxt
    ok = tmp_path / "ok.prototxt"; write_deploy_prototxt(str(ok))
    assert run("--check-prototxt", str(ok)).stdout.strip() == "ok"
    write_deploy_prototxt(str(ok), v1=True)
    assert run("--check-prototxt", str(ok)).stdout.strip() == "ok"
    write_deploy_prototxt(str(ok), extra_tail=False)                       # a file that stops at relu5_1 is enough
    assert run("--check-prototxt", str(ok)).stdout.strip() == "ok"
    # spellings the reference's own Caffe accepts for the same network: the input as an `Input` layer (input_layer.cpp), kernel_h / kernel_w / pad_h / pad_w,
    # ReLUs that write a blob of their own
    for kw in (dict(input_layer=True), dict(per_axis=True), dict(relu_in_place=False), dict(input_layer=True, per_axis=True, relu_in_place=False)):
        write_deploy_prototxt(str(ok), **kw)
        r = run("--check-prototxt", str(ok))
        assert r.stdout.strip() == "ok", (kw, r.stdout)
    bad = tmp_path / "nonsquare.prototxt"; write_deploy_prototxt(str(bad), kernel_hw={"conv2_2": (3, 5)})
    r = run("--check-prototxt", str(bad))
    assert r.returncode == 1 and "non-square kernel" in r.stdout
    for kw, needle in ((dict(drop="conv3_3"), "unexpected ReLU 'relu3_3'"), (dict(num_output={"conv2_1": 96}), "conv2_1 is num_output 96"),
                       (dict(drop="pool2"), "no 2x2 max pool after conv2_2"), (dict(drop="relu4_2"), "conv4_2 is not followed by a ReLU")):
        bad = tmp_path / "bad.prototxt"; write_deploy_prototxt(str(bad), **kw)
        r = run("--check-prototxt", str(bad))
        assert r.returncode == 1 and needle in r.stdout, (kw, r.stdout)
    (tmp_path / "junk.prototxt").write_text("layer { name: \"conv1_1\" type: \"Convolution\" ")      # unterminated message
    r = run("--check-prototxt", str(tmp_path / "junk.prototxt"))
    assert r.returncode == 1 and "not protobuf text format" in r.stdout
    r = run("--check-prototxt", str(tmp_path / "absent.prototxt"))
    assert r.returncode == 1 and "cannot open" in r.stdout


@pytest.mark.skipif(not os.path.isfile("/root/reference/demo/model/vgg19/VGG_ILSVRC_19_layers_deploy.prototxt"), reason="reference model dir not mounted")
def test_deploy_prototxt_check_accepts_the_reference_file():
    assert run("--check-prototxt", "/root/reference/demo/model/vgg19/VGG_ILSVRC_19_layers_deploy.prototxt").stdout.strip() == "ok"


@pytest.mark.gpu
def test_cli_batch_matches_library(tmp_path, ctx):
    from caffemodel_io import synthetic_vgg19, write_caffemodel
    ws, bs = synthetic_vgg19(19)
    (tmp_path / "model" / "vgg19").mkdir(parents=True)
    write_caffemodel(str(tmp_path / "model" / "vgg19" / "VGG_ILSVRC_19_layers.caffemodel"), ws, bs)
    inp = tmp_path / "in"; (inp / "sub").mkdir(parents=True)
    a, b, c = synth.image(1, 72, 64), synth.image(2, 60, 80), synth.image(3, 64, 64)
    Image.fromarray(a[..., ::-1].copy()).save(inp / "a.png"); Image.fromarray(b[..., ::-1].copy()).save(inp / "sub" / "b.png")
    Image.fromarray(c[..., ::-1].copy()).save(inp / "c.png")
    (inp / "pairs.txt").write_text("a.png sub/b.png 2.0\nmissing.png c.png 1.0\nc.png a.png 0.5\n")
    out = tmp_path / "out"
    r = run("-m", str(tmp_path / "model"), "-i", str(inp), "-o", str(out), "-g", "0")
    assert r.returncode == 0, r.stdout + r.stderr
    assert "Fail reading content image" in r.stdout                       # unreadable image: message, skip, continue (main.cu:484-496)
    assert "Patch Match Time:" in r.stdout and "**Finished Time:" in r.stdout and "Final output file:" in r.stdout
    assert r.stdout.count("Nonlocal Solve Time: ") == 10 and r.stdout.count("WLS Solve Time: ") == 10     # per level, like ColorTransfer.cpp:1373,1434
    assert sorted(os.listdir(out)) == ["a_b_2.00.png", "c_a_0.50.png", "status.jsonl"]     # <srcbase>_<refbase>_<%2.2f bds>.png (main.cu:537)
    ctx.vgg19_load_raw(ws, bs)
    for name, (s, rf, w) in {"a_b_2.00.png": (a, b, 2.0), "c_a_0.50.png": (c, a, 0.5)}.items():
        prm = nct.Params.default(); prm.bds_weight = w
        exp = ctx.process_pair(s, rf, prm)
        got = np.asarray(Image.open(out / name).convert("RGB"))[..., ::-1]
        assert np.array_equal(got, exp)


@pytest.mark.gpu
def test_cli_on_the_reference_demo_inputs(tmp_path):
    """The reference's own demo batch through the console driver: `demo/example/pairs.txt`'s layout (an `in/` sub-directory, RGBA and RGB PNGs of unequal sizes, the bds
    sweep on one pair) with the demo inputs (staged into tests/golden/natural/*.png by tests/natural_inputs.py) and the synthetic VGG19; every output PNG must carry the CRC the CPU oracle produced for that
    line (tests/golden/natural/pair_<case>.npz) — i.e. host PNG decode, alpha drop, naming, scheduling over two workers and PNG encode add nothing to the library's result."""
    import zlib, shutil, glob
    from caffemodel_io import synthetic_vgg19, write_caffemodel
    import natural_inputs
    nat = natural_inputs.require()
    cases = sorted(os.path.basename(f)[5:-4] for f in glob.glob(os.path.join(nat, "pair_*.npz")))
    if not cases:
        pytest.skip("natural fixtures not generated")
    ws, bs = synthetic_vgg19(19)
    (tmp_path / "model" / "vgg19").mkdir(parents=True)
    write_caffemodel(str(tmp_path / "model" / "vgg19" / "VGG_ILSVRC_19_layers.caffemodel"), ws, bs)
    inp = tmp_path / "example"; (inp / "in").mkdir(parents=True)
    lines = []
    for c in cases:
        a, b, bds = c.split("_")
        for n in (a, b):
            shutil.copy(os.path.join(nat, n + ".png"), inp / "in" / (n + ".png"))
        lines.append(f"in/{a}.png in/{b}.png {float(bds):.1f}")
    (inp / "pairs.txt").write_text("\n".join(lines) + "\n")
    out = tmp_path / "res"
    r = run("-m", str(tmp_path / "model"), "-i", str(inp), "-o", str(out), "-g", "0", "-inflight", "2")
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    for c in cases:
        a, b, bds = c.split("_")
        g = np.load(os.path.join(nat, f"pair_{c}.npz"))
        got = np.ascontiguousarray(np.asarray(Image.open(out / f"{a}_{b}_{float(bds):.2f}.png").convert("RGB"))[..., ::-1])
        assert zlib.crc32(got.tobytes()) == int(g["crc_canonical"]), c


@pytest.mark.gpu
def test_cli_inflight_workers_give_identical_files(tmp_path):
    """`-inflight K`: K contexts + host threads per GPU take pairs from a shared counter; the files must not depend on K."""
    from caffemodel_io import synthetic_vgg19, write_caffemodel
    ws, bs = synthetic_vgg19(19)
    (tmp_path / "model" / "vgg19").mkdir(parents=True)
    write_caffemodel(str(tmp_path / "model" / "vgg19" / "VGG_ILSVRC_19_layers.caffemodel"), ws, bs)
    inp = tmp_path / "in"; inp.mkdir()
    lines = []
    for i, (h, w) in enumerate([(64, 64), (72, 56), (48, 80), (96, 64), (64, 96)]):
        Image.fromarray(synth.image(10 + i, h, w)[..., ::-1].copy()).save(inp / f"s{i}.png")
        Image.fromarray(synth.image(20 + i, w, h)[..., ::-1].copy()).save(inp / f"r{i}.png")
        lines.append(f"s{i}.png r{i}.png 2.0\n")
    (inp / "pairs.txt").write_text("".join(lines))
    outs = {}
    for k in (1, 3):
        out = tmp_path / f"out{k}"
        r = run("-m", str(tmp_path / "model"), "-i", str(inp), "-o", str(out), "-g", "0", "-inflight", str(k))
        assert r.returncode == 0, r.stdout + r.stderr
        assert f"{k} in flight each" in r.stdout
        outs[k] = {n: np.asarray(Image.open(out / n)) for n in sorted(os.listdir(out)) if n.endswith(".png")}
    assert list(outs[1]) == list(outs[3]) and len(outs[1]) == 5
    for n in outs[1]:
        assert np.array_equal(outs[1][n], outs[3][n]), n


@pytest.mark.gpu
def test_cli_process_per_gpu_shapes_give_identical_files(tmp_path):
    """the process-per-GPU shape of the driver. `-procs 2` forks two processes (rank r of 2, device -g + r — both mapped to device 0 here through
    NCT_DEVICE_OVERRIDE) that share pairs.txt by i mod 2 and write status.<r>.jsonl; `-world 2 -rank r -steal 1` started by hand draws lines from <out>/.tickets under a
    file lock. Every shape must leave the files of the one-process run, each pair exactly once."""
    import json
    from caffemodel_io import synthetic_vgg19, write_caffemodel
    ws, bs = synthetic_vgg19(19)
    (tmp_path / "model" / "vgg19").mkdir(parents=True)
    write_caffemodel(str(tmp_path / "model" / "vgg19" / "VGG_ILSVRC_19_layers.caffemodel"), ws, bs)
    inp = tmp_path / "in"; inp.mkdir()
    lines = []
    for i, (h, w) in enumerate([(64, 64), (72, 56), (48, 80), (96, 64), (64, 96)]):
        Image.fromarray(synth.image(30 + i, h, w)[..., ::-1].copy()).save(inp / f"s{i}.png")
        Image.fromarray(synth.image(40 + i, w, h)[..., ::-1].copy()).save(inp / f"r{i}.png")
        lines.append(f"s{i}.png r{i}.png 2.0\n")
    (inp / "pairs.txt").write_text("".join(lines))
    common = ["-m", str(tmp_path / "model"), "-i", str(inp)]
    env = dict(os.environ, NCT_DEVICE_OVERRIDE="0")

    def pngs(d):
        return {n: np.asarray(Image.open(d / n)) for n in sorted(os.listdir(d)) if n.endswith(".png")}

    def status(d, files):
        recs = []
        for f in files:
            if os.path.exists(d / f):            # a rank that drew no line (-steal: the other one was faster) writes no file
                recs += [json.loads(l) for l in open(d / f)]
        return recs
    r = run(*common, "-o", str(tmp_path / "one"), "-g", "0")
    assert r.returncode == 0, r.stdout + r.stderr
    ref = pngs(tmp_path / "one")
    assert len(ref) == 5
    # -procs 2: static split
    r = subprocess.run([BIN, *common, "-o", str(tmp_path / "procs"), "-g", "0", "-procs", "2"], capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "Rank 0 of 2: Processed 3 pair(s)" in r.stdout and "Rank 1 of 2: Processed 2 pair(s)" in r.stdout and "All 2 process(es) finished" in r.stdout
    got = pngs(tmp_path / "procs")
    assert list(got) == list(ref) and all(np.array_equal(got[n], ref[n]) for n in ref)
    recs = status(tmp_path / "procs", ["status.0.jsonl", "status.1.jsonl"])
    assert sorted(x["pair"] for x in recs) == [0, 1, 2, 3, 4] and all(x["status"] == "done" for x in recs)
    assert not os.path.exists(tmp_path / "procs" / "status.jsonl")
    # -procs 2 -steal 1: the shared counter
    r = subprocess.run([BIN, *common, "-o", str(tmp_path / "steal"), "-g", "0", "-procs", "2", "-steal", "1", "-inflight", "2"], capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    got = pngs(tmp_path / "steal")
    assert list(got) == list(ref) and all(np.array_equal(got[n], ref[n]) for n in ref)
    recs = status(tmp_path / "steal", ["status.0.jsonl", "status.1.jsonl"])
    assert sorted(x["pair"] for x in recs) == [0, 1, 2, 3, 4]
    # two ranks started by hand, one after the other: the second finds the counter where the first left it (nothing left) — each pair still exactly once
    out = tmp_path / "hand"
    r0 = subprocess.run([BIN, *common, "-o", str(out), "-g", "0", "-world", "2", "-rank", "0", "-steal", "1"], capture_output=True, text=True, env=env)
    r1 = subprocess.run([BIN, *common, "-o", str(out), "-g", "0", "-world", "2", "-rank", "1", "-steal", "1"], capture_output=True, text=True, env=env)
    assert r0.returncode == 0 and r1.returncode == 0, r0.stdout[-2000:] + r1.stdout[-2000:]
    assert "Rank 0 of 2: Processed 5 pair(s)" in r0.stdout and "Rank 1 of 2: Processed 0 pair(s)" in r1.stdout
    got = pngs(out)
    assert list(got) == list(ref) and all(np.array_equal(got[n], ref[n]) for n in ref)
    # a rank outside the world is refused
    r = run(*common, "-o", str(tmp_path / "bad"), "-world", "2", "-rank", "2")
    assert r.returncode == 255 and "is not in [0, -world 2)" in r.stdout
    # -rccl 1 (round 6): the ranks form an RCCL communicator for the start barrier and the MAX / SUM reductions of the job's time and pair count. One rank on this 1-GPU box
    # executes the whole code path (id file, ncclCommInitRank, three all-reduces, destroy); the files do not depend on it
    r = subprocess.run([BIN, *common, "-o", str(tmp_path / "rccl1"), "-g", "0", "-world", "1", "-rccl", "1"], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "RCCL: rank 0 of 1 joined, start barrier passed." in r.stdout and "All 1 rank(s) over RCCL: 5 pair(s) in " in r.stdout, r.stdout[-2000:]
    got = pngs(tmp_path / "rccl1")
    assert list(got) == list(ref) and all(np.array_equal(got[n], ref[n]) for n in ref)
    assert not os.path.exists(tmp_path / "rccl1" / ".rccl_id")
    # two ranks on ONE device: RCCL refuses a communicator with a duplicate GPU — every rank says so and goes on without the barrier; same files, each pair once
    r = subprocess.run([BIN, *common, "-o", str(tmp_path / "rccl2"), "-g", "0", "-procs", "2", "-rccl", "1"], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert r.stdout.count("goes on without the barrier") == 2 or r.stdout.count("start barrier passed") == 2, r.stdout[-3000:]
    got = pngs(tmp_path / "rccl2")
    assert list(got) == list(ref) and all(np.array_equal(got[n], ref[n]) for n in ref)


@pytest.mark.gpu
def test_cli_shrink_jpeg_resume_levels(tmp_path, ctx):
    """transfer_single's shrink-to-1000 path (main.cu:500-522: int truncation of the shorter side) on a JPEG input, `-levels 1`
    (BASELINE config 1), and `-resume 1` (existing outputs are skipped; status.jsonl gets a line per pair)."""
    import json
    from caffemodel_io import synthetic_vgg19, write_caffemodel
    ws, bs = synthetic_vgg19(19)
    (tmp_path / "model" / "vgg19").mkdir(parents=True)
    write_caffemodel(str(tmp_path / "model" / "vgg19" / "VGG_ILSVRC_19_layers.caffemodel"), ws, bs)
    inp = tmp_path / "in"; inp.mkdir()
    big = synth.image(4, 1100, 700)                                            # 1100 high: shrinks to 1000 x (int)(1000/1100*700) = 636
    Image.fromarray(big[..., ::-1].copy()).save(inp / "big.jpg", quality=90, subsampling=2)
    small = synth.image(5, 120, 160)
    Image.fromarray(small[..., ::-1].copy()).save(inp / "small.png")
    (inp / "pairs.txt").write_text("big.jpg small.png 2.0\nsmall.png big.jpg 1.0\n")
    out = tmp_path / "out"
    args = ("-m", str(tmp_path / "model"), "-i", str(inp), "-o", str(out), "-levels", "1")
    r = run(*args)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "w = 700, h = 1100" in r.stdout
    assert sorted(n for n in os.listdir(out) if n.endswith(".png")) == ["big_small_2.00.png", "small_big_1.00.png"]
    got = np.asarray(Image.open(out / "big_small_2.00.png").convert("RGB"))[..., ::-1]
    assert got.shape == (1000, 636, 3)
    # same pixels as the library fed with the decoded + shrunk images
    dec = np.asarray(Image.open(inp / "big.jpg").convert("RGB"))[..., ::-1]
    ctx.vgg19_load_raw(ws, bs)
    prm = nct.Params.default(); prm.levels = 1
    exp = ctx.process_pair(ctx.resize_u8c3(dec, 1000, 636), small, prm)
    assert np.array_equal(got, exp)
    st = [json.loads(l) for l in (out / "status.jsonl").read_text().splitlines()]
    assert [s["status"] for s in st] == ["done", "done"]            # (the I/O pool finishes pairs in any order: look pairs up by index)
    assert {s["pair"]: os.path.basename(s["output"]) for s in st} == {0: "big_small_2.00.png", 1: "small_big_1.00.png"}
    mtime = os.path.getmtime(out / "big_small_2.00.png")
    os.remove(out / "small_big_1.00.png")
    r = run(*args, "-resume", "1")
    assert r.returncode == 0 and "Skipping (-resume)" in r.stdout
    assert os.path.getmtime(out / "big_small_2.00.png") == mtime and os.path.exists(out / "small_big_1.00.png")
    st = [json.loads(l) for l in (out / "status.jsonl").read_text().splitlines()]
    assert sorted(s["status"] for s in st[2:]) == ["done", "skipped"]


@pytest.mark.gpu
def test_cli_vis_dumps(tmp_path, ctx, oracle):
    """`-vis 1` (the reference's ENABLE_VIS, Config.h:8): per level <pre>_aFlow/_bFlow/_tCnt/_tStl/_errMap_<l>.png with the reference's
    arithmetic (reconstruct_flow GeneralizedPatchMatch.cu:337-353; getHeat ColorTransfer.cpp:1127-1178), plus guide_/result_; the
    final output is unchanged by the flag."""
    from caffemodel_io import synthetic_vgg19, write_caffemodel
    ws, bs = synthetic_vgg19(19)
    (tmp_path / "model" / "vgg19").mkdir(parents=True)
    write_caffemodel(str(tmp_path / "model" / "vgg19" / "VGG_ILSVRC_19_layers.caffemodel"), ws, bs)
    inp = tmp_path / "in"; inp.mkdir()
    a, b = synth.image(1, 80, 64), synth.image(2, 64, 96)
    Image.fromarray(a[..., ::-1].copy()).save(inp / "a.png"); Image.fromarray(b[..., ::-1].copy()).save(inp / "b.png")
    (inp / "pairs.txt").write_text("a.png b.png 2.0\n")
    out = tmp_path / "out"
    r = run("-m", str(tmp_path / "model"), "-i", str(inp), "-o", str(out), "-vis", "1")
    assert r.returncode == 0, r.stdout + r.stderr
    names = set(os.listdir(out))
    for l in range(5):
        for what in ("aFlow", "bFlow", "tCnt", "tStl", "errMap", "guide", "result", "aVis", "bVis", "aVis_init", "bVis_init", "refine_init",
                     "aVis_nonlocal", "bVis_nonlocal", "refine_nonlocal", "knn", "patchVis"):
            assert f"a_b_2.00_{what}_{l}.png" in names, (what, l)
    assert "a_b_2.00_cluster_small.png" in names
    ctx.vgg19_load_raw(ws, bs)
    ctx.pair_upload(a, b)
    lv = ctx.pair_run_levels(a.shape, b.shape, want_color=True)
    load = lambda n: np.asarray(Image.open(out / n).convert("RGB"))[..., ::-1]
    assert np.array_equal(load("a_b_2.00.png"), ctx.pair_download())
    assert np.array_equal(load("a_b_2.00_result_4.png"), lv["result"][4]) and np.array_equal(load("a_b_2.00_guide_2.png"), lv["guide"][2])
    assert np.array_equal(load("a_b_2.00_tCnt_4.png"), a) and np.array_equal(load("a_b_2.00_tStl_3.png"), oracle.resize_u8c3(b, 32, 48))
    ann = lv["ann"][3]; bh, bw = lv["dims"][3][2:]
    flow = load("a_b_2.00_aFlow_3.png")
    assert np.array_equal(flow[..., 0], (255 * ((ann & 0xFFF).astype(np.float32) / np.float32(bw))).astype(np.uint8))
    assert np.array_equal(flow[..., 2], (255 * ((ann >> 12).astype(np.float32) / np.float32(bh))).astype(np.uint8)) and not flow[..., 1].any()
    hm = load("a_b_2.00_errMap_0.png")
    e = lv["err"][0].astype(np.float64); i = np.unravel_index(np.argmin(e), e.shape); j = np.unravel_index(np.argmax(e), e.shape)
    assert hm[i].tolist() == [128, 0, 0] and hm[j].tolist() == [0, 0, 128]          # getHeat(0) = dark blue, getHeat(1) = dark red (BGR)
    # coefficient images (ColorTransfer.cpp:1286-1296, 1400-1410, 1451-1462): int(a*50), int(b*255+127) clamped; recoloured source
    H, W = a.shape[:2]
    byte = lambda v: np.clip(v, 0, 255).astype(np.int64).astype(np.uint8)
    for l in (0, 2, 4):
        col = lv["color"][l]
        ab = col["ab_wls"].reshape(2, H, W, 3)
        assert np.array_equal(load(f"a_b_2.00_aVis_{l}.png"), byte(ab[0] * 50)) and np.array_equal(load(f"a_b_2.00_bVis_{l}.png"), byte(ab[1] * 255 + 127))
        ah, aw = lv["dims"][l][:2]; smp = 1 << (4 - l)
        yy, xx = np.meshgrid(np.arange(H) // smp, np.arange(W) // smp, indexing="ij")
        loc = col["ab_local"].reshape(2, ah, aw, 3)[:, yy, xx]
        assert np.array_equal(load(f"a_b_2.00_aVis_init_{l}.png"), byte(loc[0] * 50)) and np.array_equal(load(f"a_b_2.00_bVis_init_{l}.png"), byte(loc[1] * 255 + 127))
        lab = oracle.bgr2lab(a).astype(np.float64) / 255.0
        up = col["ab_up"].reshape(2, H, W, 3)
        for tag, cf in (("init", loc), ("nonlocal", up)):
            rec = np.rint(np.clip(lab * cf[0] + cf[1], 0.0, 1.0) * 255.0).astype(np.uint8)
            assert np.array_equal(load(f"a_b_2.00_refine_{tag}_{l}.png"), oracle.lab2bgr(rec)), (tag, l)
    # the S2 coefficients recolour the source into the level's intermediate result
    ab = lv["color"][4]["ab_wls"].reshape(2, H, W, 3)
    rec = np.rint(np.clip(oracle.bgr2lab(a).astype(np.float64) / 255.0 * ab[0] + ab[1], 0.0, 1.0) * 255.0).astype(np.uint8)
    assert np.array_equal(oracle.lab2bgr(rec), lv["result"][4])
    # patchVis (ColorTransfer.cpp:1190-1221): per level pixel the clipped 3x3 window of the guidance above that of the level image
    pv = load("a_b_2.00_patchVis_1.png"); ah, aw = lv["dims"][1][:2]; g = lv["guide"][1]; c1 = load("a_b_2.00_tCnt_1.png")
    assert pv.shape == (ah * 6, aw * 3, 3)
    for (y, x) in ((0, 0), (3, 5), (ah - 1, aw - 1), (0, aw - 1)):
        y0, x0, y1, x1 = max(y - 1, 0), max(x - 1, 0), min(y + 2, ah), min(x + 2, aw)
        cell = np.zeros((6, 3, 3), np.uint8); cell[:y1 - y0, :x1 - x0] = g[y0:y1, x0:x1]; cell[3:3 + y1 - y0, :x1 - x0] = c1[y0:y1, x0:x1]
        assert np.array_equal(pv[6 * y:6 * y + 6, 3 * x:3 * x + 3], cell), (y, x)
    # cluster images: one colour per k-means label; knn_<l> = the same labels seen from level l (x / 2^l)
    cl = load("a_b_2.00_cluster_small.png"); lab0 = lv["labels"]
    assert cl.shape[:2] == lab0.shape
    for k in np.unique(lab0):
        assert len(np.unique(cl[lab0 == k].reshape(-1, 3), axis=0)) == 1
    assert len(np.unique(cl.reshape(-1, 3), axis=0)) == len(np.unique(lab0))
    kn = load("a_b_2.00_knn_2.png"); ah, aw = lv["dims"][2][:2]
    yy, xx = np.meshgrid(np.minimum(np.arange(ah) // 4, lab0.shape[0] - 1), np.minimum(np.arange(aw) // 4, lab0.shape[1] - 1), indexing="ij")
    assert np.array_equal(kn, cl[yy, xx])


@pytest.mark.gpu
def test_cli_two_logical_gpus_io_pool_shared_weights(tmp_path, ctx):
    """The 8-GPU host path on one GPU: `-gpus 2 -inflight 2` with NCT_DEVICE_OVERRIDE=0 (every logical GPU on device 0) — four worker contexts, ONE parse of
    the caffemodel and ONE device copy of the weights shared by all of them, pairs decoded/encoded by the I/O pool (`-io 3`), the prototxt read and checked;
    the files equal `-io 0` (workers doing their own I/O) and the library, whatever worker ran a pair. A prototxt describing another net is refused."""
    from caffemodel_io import synthetic_vgg19, write_caffemodel, write_deploy_prototxt
    ws, bs = synthetic_vgg19(19)
    mdir = tmp_path / "model" / "vgg19"; mdir.mkdir(parents=True)
    write_caffemodel(str(mdir / "VGG_ILSVRC_19_layers.caffemodel"), ws, bs)
    write_deploy_prototxt(str(mdir / "VGG_ILSVRC_19_layers_deploy.prototxt"))
    inp = tmp_path / "in"; inp.mkdir()
    imgs, lines = {}, []
    for i, (h, w) in enumerate([(64, 64), (72, 56), (48, 80), (96, 64), (64, 96), (80, 80), (56, 72)]):
        imgs[i] = (synth.image(30 + i, h, w), synth.image(40 + i, w, h))
        Image.fromarray(imgs[i][0][..., ::-1].copy()).save(inp / f"s{i}.png"); Image.fromarray(imgs[i][1][..., ::-1].copy()).save(inp / f"r{i}.png")
        lines.append(f"s{i}.png r{i}.png 2.0\n")
    (inp / "pairs.txt").write_text("".join(lines))
    env = dict(os.environ, NCT_DEVICE_OVERRIDE="0")
    outs = {}
    for io in ("3", "0"):
        out = tmp_path / f"out_io{io}"
        r = subprocess.run([BIN, "-m", str(tmp_path / "model"), "-i", str(inp), "-o", str(out), "-gpus", "2", "-inflight", "2", "-io", io], capture_output=True, text=True, env=env)
        assert r.returncode == 0, r.stdout + r.stderr
        assert "VGG19 weights: parsed once, 1 device copy of" in r.stdout and "shared by 4 context(s)" in r.stdout
        assert f"on 2 GPU(s), 2 in flight each, {io} I/O thread(s)" in r.stdout and "not found" not in r.stdout
        assert r.stdout.count("Final output file:") == 7 and not [n for n in os.listdir(out) if n.endswith(".tmp")]
        outs[io] = {n: np.asarray(Image.open(out / n).convert("RGB"))[..., ::-1] for n in sorted(os.listdir(out)) if n.endswith(".png")}
    assert list(outs["3"]) == list(outs["0"]) and len(outs["3"]) == 7
    ctx.vgg19_load_raw(ws, bs)
    for i in range(7):
        n = f"s{i}_r{i}_2.00.png"
        assert np.array_equal(outs["3"][n], outs["0"][n]), n
        if i < 3:
            assert np.array_equal(outs["3"][n], ctx.process_pair(*imgs[i])), n
    # -resume: a truncated output (killed run) is redone, a complete one skipped
    out = tmp_path / "out_io3"
    full = (out / "s0_r0_2.00.png").read_bytes()
    (out / "s1_r1_2.00.png").write_bytes(full[: len(full) // 2])
    r = subprocess.run([BIN, "-m", str(tmp_path / "model"), "-i", str(inp), "-o", str(out), "-resume", "1"], capture_output=True, text=True, env=env)
    assert r.returncode == 0 and r.stdout.count("Skipping (-resume)") == 6 and r.stdout.count("Final output file:") == 1
    assert np.array_equal(np.asarray(Image.open(out / "s1_r1_2.00.png").convert("RGB"))[..., ::-1], outs["0"]["s1_r1_2.00.png"])
    write_deploy_prototxt(str(mdir / "VGG_ILSVRC_19_layers_deploy.prototxt"), num_output={"conv4_1": 256})
    r = subprocess.run([BIN, "-m", str(tmp_path / "model"), "-i", str(inp), "-o", str(tmp_path / "o2")], capture_output=True, text=True, env=env)
    assert r.returncode == 255 and "conv4_1 is num_output 256" in r.stdout


@pytest.mark.gpu
def test_cli_eight_logical_gpus_four_in_flight(tmp_path):
    """The full-node host shape in ONE process on one device: `-gpus 8 -inflight 4` with NCT_DEVICE_OVERRIDE=0 = 32 worker contexts (arena, streams,
    host thread each) + 16 I/O threads, one parse and ONE device copy of the weights. Every pair is processed exactly once, the files are byte-identical to `-gpus 1
    -inflight 2`, also with the decoded backlog squeezed to a single pair (NCT_IO_READY_MB=0: the byte bound of the ready queue), and duplicate pairs.txt lines (same
    output name, encoded concurrently) leave no temporary file behind. scripts/cli_8gpu_shape.py measures the throughput of this shape on 700x700 pairs."""
    from caffemodel_io import synthetic_vgg19, write_caffemodel, write_deploy_prototxt
    ws, bs = synthetic_vgg19(19)
    mdir = tmp_path / "model" / "vgg19"; mdir.mkdir(parents=True)
    write_caffemodel(str(mdir / "VGG_ILSVRC_19_layers.caffemodel"), ws, bs)
    write_deploy_prototxt(str(mdir / "VGG_ILSVRC_19_layers_deploy.prototxt"), input_layer=True)
    inp = tmp_path / "in"; inp.mkdir()
    lines = []
    for i in range(40):
        h, w = 64 + 8 * (i % 5), 64 + 8 * (i % 3)
        Image.fromarray(synth.image(300 + i, h, w)[..., ::-1].copy()).save(inp / f"s{i}.png"); Image.fromarray(synth.image(400 + i, w, h)[..., ::-1].copy()).save(inp / f"r{i}.png")
        lines.append(f"s{i}.png r{i}.png 2.0\n")
    lines += lines[:6]                                              # six duplicate lines: two writers of the same output name
    (inp / "pairs.txt").write_text("".join(lines))
    blobs = {}
    for tag, args, extra in (("ref", ["-gpus", "1", "-inflight", "2"], {}), ("node", ["-gpus", "8", "-inflight", "4"], {}), ("squeezed", ["-gpus", "8", "-inflight", "4"], {"NCT_IO_READY_MB": "0"})):
        out = tmp_path / f"out_{tag}"
        r = subprocess.run([BIN, "-m", str(tmp_path / "model"), "-i", str(inp), "-o", str(out), *args], capture_output=True, text=True, env=dict(os.environ, NCT_DEVICE_OVERRIDE="0", **extra))
        assert r.returncode == 0, r.stdout[-3000:] + r.stderr
        if tag != "ref":
            assert "1 device copy of" in r.stdout and "shared by 32 context(s)" in r.stdout and "on 8 GPU(s), 4 in flight each" in r.stdout
        assert r.stdout.count("Final output file:") == 46
        names = sorted(os.listdir(out))
        assert not [n for n in names if ".tmp" in n] and len([n for n in names if n.endswith(".png")]) == 40
        blobs[tag] = {n: (out / n).read_bytes() for n in names if n.endswith(".png")}
    assert blobs["node"] == blobs["ref"] and blobs["squeezed"] == blobs["ref"]


@pytest.mark.gpu
def test_contexts_share_one_weight_copy(ctx, tmp_path):
    """nct_model_parse_caffemodel + nct_vgg19_load_model + nct_vgg19_share_weights: N contexts on one GPU hold ONE device copy (same address, sharers = N), a
    sharing context computes the same features, reloading into a sharer detaches it, and the copy outlives the context that uploaded it."""
    from caffemodel_io import synthetic_vgg19, write_caffemodel
    ws, bs = synthetic_vgg19(19)
    path = str(tmp_path / "m.caffemodel"); write_caffemodel(path, ws, bs, fmt="v1")
    model = nct.Model(path)
    a = nct.Context(0); a.vgg19_load_model(model); model.close()
    others = [nct.Context(0) for _ in range(3)]
    for o in others:
        o.vgg19_share_weights(a)
    ia = a.vgg19_weights_info()
    assert ia["sharers"] == 4 and 45e6 < ia["bytes"] < 60e6
    assert all(o.vgg19_weights_info()["id"] == ia["id"] for o in others)
    img = synth.image(9, 40, 48)
    fa = a.vgg19_features(img, 5)
    a.close()                                                       # the uploader goes away: the copy stays with its remaining users
    assert others[0].vgg19_weights_info()["sharers"] == 3
    fo = others[0].vgg19_features(img, 5)
    assert all(np.array_equal(x.view(np.uint32), y.view(np.uint32)) for x, y in zip(fa, fo))
    others[1].vgg19_load_raw(ws, bs)                                # a reload never writes into a shared copy
    assert others[1].vgg19_weights_info()["id"] != ia["id"] and others[0].vgg19_weights_info()["sharers"] == 2
    with pytest.raises(nct.NctError):
        nct.Context(0).vgg19_share_weights(nct.Context(0))          # nothing loaded in the source
    with pytest.raises(nct.NctError):
        nct.Model(str(tmp_path / "absent.caffemodel"))
    for o in others:
        o.close()


def _parse_only(argv):
    r = run("--parse-only", *argv)
    assert r.returncode == 0, r.stdout
    head, tail = r.stdout.split("@@RESULT ", 1)
    lines = tail.strip().split("\n")
    vals = dict(l.split("=", 1) for l in lines[1:])
    values = {"m": vals["m"], "i": vals["i"], "o": vals["o"], "g": int(vals["g"]), "files": int(vals["files"]),
              **{k: float(vals[k]) for k in ("bds", "eps", "nl", "l", "w")}}
    return int(lines[0].split("=")[1]), head.split("\n"), values


def test_cli_parser_matches_the_reference_cmdline():
    """D1 pinned by the reference's own parser: tests/golden/cmdline_ref.json was written by oracle/_ref/ref_cmdline = the reference's CmdLine.cpp + CmdLine.h
    compiled unmodified with get_input's registrations (main.cu:29-44; generator tests/golden/gen_cmdline_ref.py). For every recorded argument vector the product
    CLI must return the same verdict (parsed / `return -1`), leave the same values in the nine parameters, count the same positional tokens and print the same
    text: "Unrecognized parameter: …", and the help lines of the nine reference parameters with the values parsed so far as "(default=…)" — the product's
    [extension] flags follow them. The two portable extensions (negative numbers, unix paths as values) are the cases the reference cannot take at all; the
    fixture records its refusal and the product's specified result."""
    import json
    fx = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "cmdline_ref.json")))
    assert len(fx["cases"]) >= 30
    n_ext = 0
    for c in fx["cases"]:
        rc, printed, values = _parse_only(c["argv"])
        if "extension" in c:
            n_ext += 1
            assert c["rc"] == 0 and rc == c["extension"]["rc"], c["argv"]
            for k, v in c["extension"].items():
                if k != "rc":
                    assert values[k] == v, (c["argv"], k)
            continue
        assert rc == c["rc"], c["argv"]
        assert values == c["values"], (c["argv"], values, c["values"])
        ref = [l for l in c["printed"] if not l.startswith("Running: ")]
        got = [l for l in printed if not l.startswith("Running: ") and "[extension]" not in l]
        assert got == ref, (c["argv"], got, ref)
        assert sum(l.startswith("Running: ") for l in printed) == sum(l.startswith("Running: ") for l in c["printed"])
    assert n_ext == 3


@pytest.mark.skipif(not os.path.exists(os.path.join(os.path.dirname(nct.PKG_ROOT), "oracle", "_ref", "ref_cmdline")), reason="oracle/_ref not built (reference not mounted)")
def test_cmdline_fixture_is_what_the_reference_parser_prints_now():
    """Where oracle/_ref exists (the build container), the committed fixture is re-derived live for a few vectors, so it cannot drift from the generator."""
    import json
    exe = os.path.join(os.path.dirname(nct.PKG_ROOT), "oracle", "_ref", "ref_cmdline")
    fx = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "cmdline_ref.json")))
    for c in fx["cases"][::4]:
        r = subprocess.run([exe] + c["argv"], capture_output=True, text=True, check=True)
        head, tail = r.stdout.split("@@RESULT ", 1)
        assert int(tail.split("\n")[0].split("=")[1]) == c["rc"]
        assert [l for l in head.split("\n") if not l.startswith("Running: ")] == [l for l in c["printed"] if not l.startswith("Running: ")]
