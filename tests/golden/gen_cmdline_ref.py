#!/usr/bin/env python3
"""Generates tests/golden/cmdline_ref.json with the reference's OWN command-line parser: oracle/_ref/ref_cmdline (built by
`make -C oracle _ref` from oracle/ref_cmdline.cpp + /root/reference/.../source/CmdLine.cpp and CmdLine.h, compiled unmodified from
where they lie, with get_input's registrations main.cu:29-44 over Config::Config()'s defaults Config.h:58-72) is run on the argument
vectors below. The fixture holds the inputs (argv) and the outputs (what Parse returned, what it printed, the values main would go on
with); tests/test_cli.py::test_cli_parser_matches_the_reference_cmdline runs the product CLI's `--parse-only` hook on the same vectors.
`extension` marks the two documented cases where the product deliberately does something else than the reference:
the fixture then records BOTH the reference's behaviour and the product's specified one. Needs /root/reference (this container only)."""
import json, os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
exe = os.path.join(REPO, "oracle", "_ref", "ref_cmdline")
subprocess.run(["make", "-C", os.path.join(REPO, "oracle"), "_ref"], check=True)
assert os.path.exists(exe), "oracle/_ref/ref_cmdline missing (is /root/reference mounted?)"

VECTORS = [
    [],
    ["-m", "models", "-i", "in", "-o", "out"],
    ["-m", "models", "-i", "in", "-o", "out", "-g", "3", "-bds", "4.5", "-eps", "0.1", "-nl", "0.4", "-l", "0.001", "-w", "0.0234375"],
    ["/m", "models", "/i", "in", "/g", "1"],                      # the '/' form of every flag
    ["-i", "..\\example\\in\\", "-o", "..\\example\\res\\", "-m", "..\\..\\models\\"],      # the demo's run.bat shape
    ["-bds", "1e-3", "-eps", "6E1", "-w", ".5"],
    ["-g", "12abc"],                                              # operator>>: 12
    ["-g", "abc"],                                                # failed extraction: 0
    ["-g", "3.7"],                                                # 3
    ["-g", "+4"],
    ["-bds", "0x10"],                                             # operator>> stops at 'x': 0
    ["-bds", "nanx", "-eps", "7"],
    ["-bds"],                                                     # value missing at the end
    ["-g", "-bds", "3"],                                          # the next token is an argument: -g keeps its value, -bds is parsed
    ["-m", "-h"],                                                 # ... and that argument may be the help
    ["-m", "a", "-m", "b"],                                       # later wins
    ["stray", "-g", "2", "another"],                              # positional tokens are collected, not an error
    ["-g", "2", ""],                                              # an empty token is a positional
    ["-m", ""],                                                   # an empty value is not a value (TParm::Parse): unchanged, then positional
    ["-h"], ["-?"], ["-help"], ["/h"], ["/?"],
    ["-g", "5", "-h", "-bds", "9"],                               # help printed with the values parsed SO FAR as "defaults"
    ["-zzz", "1"],                                                # unknown
    ["-G", "1"],                                                  # flags are case sensitive
    ["-bds=3"],                                                   # no '=' form
    ["--m", "x"],                                                 # no double dash
    ["-"],                                                        # a bare dash is an argument with an empty name
    ["-g", "2", "-5"],                                            # a stray negative number is an (unknown) argument
    # --- the two portable extensions of the product (the reference ends both with "Unrecognized parameter")
    ["-bds", "-1.5"],
    ["-g", "-1", "-bds", "-.5"],
    ["-m", "/data/models", "-i", "/data/in", "-o", "/tmp/out"],
    ["-m", "/i", "in"],                                           # NOT the extension: "/i" names a parameter, so it is not a path
]
EXTENSION = {31: {"rc": 1, "bds": -1.5}, 32: {"rc": 1, "g": -1, "bds": -0.5}, 33: {"rc": 1, "m": "/data/models", "i": "/data/in", "o": "/tmp/out"}}


def run(argv):
    r = subprocess.run([exe] + argv, capture_output=True, text=True, check=True)
    head, tail = r.stdout.split("@@RESULT ", 1)
    lines = tail.strip().split("\n")
    vals = dict(l.split("=", 1) for l in lines[1:])
    # the first help line is "Running: <argv[0]>": keep the text after it (program names differ)
    printed = [l for l in head.split("\n")]
    return {"argv": argv, "rc": int(lines[0].split("=")[1]), "printed": printed,
            "values": {"m": vals["m"], "i": vals["i"], "o": vals["o"], "g": int(vals["g"]), "files": int(vals["files"]),
                       **{k: float(vals[k]) for k in ("bds", "eps", "nl", "l", "w")}}}


cases = []
for n, v in enumerate(VECTORS):
    c = run(v)
    if n in EXTENSION:
        assert c["rc"] == 0 and any("Unrecognized parameter" in l for l in c["printed"]), (n, v, c)   # the reference cannot take these at all
        c["extension"] = EXTENSION[n]
    cases.append(c)
json.dump({"generator": "tests/golden/gen_cmdline_ref.py", "reference": "source/CmdLine.cpp + CmdLine.h (unmodified) + get_input main.cu:29-44",
           "cases": cases}, open(os.path.join(HERE, "cmdline_ref.json"), "w"), indent=1)
print("wrote %d cases" % len(cases))
