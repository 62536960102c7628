"""Build the C oracle (test infrastructure).  `python -m oracle.build`"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "raster.c")
OUT = os.path.join(HERE, "liboracle_raster.so")


def build(force: bool = False) -> str:
    if (not force and os.path.exists(OUT)
            and os.path.getmtime(OUT) >= os.path.getmtime(SRC)):
        return OUT
    cmd = ["gcc", "-O2", "-ffp-contract=off", "-fopenmp", "-shared", "-fPIC",
           "-o", OUT, SRC, "-lm"]
    subprocess.check_call(cmd)
    return OUT


def build_ref() -> str | None:
    """Compile the reference's own kernel source for the CPU (only where /root/reference exists)."""
    script = os.path.join(HERE, "build_ref.sh")
    if not os.path.isdir("/root/reference"):
        return None
    subprocess.check_call(["bash", script])
    return os.path.join(HERE, "_ref", "libpcpr_ref.so")


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
    r = build_ref()
    print(r if r else "reference not present: oracle/_ref not rebuilt")
