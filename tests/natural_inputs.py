"""The reference's demo photographs (demo/example/in/{in,tar}{0,1,2,3,4}.png: all ten images of demo/example/pairs.txt) as test INPUTS — not committed (third-party images of unknown licence).

They are staged, byte for byte, into tests/golden/natural/ (git-ignored, but not gpurun-ignored: like oracle/_ref they travel to the GPU box with the snapshot) by
`python tests/natural_inputs.py`, which __graft_entry__.build() runs wherever a source directory exists: $NCT_DEMO_DIR, else /root/reference/demo/example/in.
What IS committed are the derived fixtures tests/golden/natural/pair_*.npz (CRCs of the oracle's results on them, incl. the CRC of the decoded inputs, so a staged file
that is not the generator's is noticed). Tests call require() and are skipped where the photographs are absent."""
import os
import shutil

NAMES = ("in0", "in1", "in2", "in3", "in4", "tar0", "tar1", "tar2", "tar3", "tar4")
DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "natural")


def source_dir():
    for d in (os.environ.get("NCT_DEMO_DIR"), "/root/reference/demo/example/in"):
        if d and all(os.path.exists(os.path.join(d, n + ".png")) for n in NAMES):
            return d
    return None


def present():
    return all(os.path.exists(os.path.join(DIR, n + ".png")) for n in NAMES)


def stage():
    """copies the ten photographs from the source directory if they are not there yet; returns True when they are present afterwards"""
    src = source_dir()
    if src and not present():
        os.makedirs(DIR, exist_ok=True)
        for n in NAMES:
            shutil.copyfile(os.path.join(src, n + ".png"), os.path.join(DIR, n + ".png"))
    return present()


def require():
    if not stage():
        import pytest
        pytest.skip("demo photographs not staged (set NCT_DEMO_DIR or run `python tests/natural_inputs.py` where /root/reference is mounted)")
    return DIR


if __name__ == "__main__":
    print("staged" if stage() else "no source directory: nothing staged", DIR)
