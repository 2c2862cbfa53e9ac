"""The compile flags of the product's two objects per length (smfft_amd/csrc/Makefile: smfft_inst_<N>.o = external kernels + dispatch,
smfft_mult_<N>.o = the in-LDS `multiple` kernels with MULT_FLAGS_<N>), for the tools and tests that compile smfft_inst.hip by themselves and
must look at what ships.
    from inst_flags import mult_flags, part_flags
    python tools/inst_flags.py 1024        (prints the flags of the in-LDS object of that length)"""
import os
import re
import sys

MAKEFILE = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "smfft_amd", "csrc", "Makefile")


def mult_flags(n):
    """extra hipcc flags of smfft_mult_<n>.o"""
    m = re.search(r"^MULT_FLAGS_%d\s*:=(.*)$" % int(n), open(MAKEFILE).read(), re.M)
    if m is None:
        raise KeyError(f"MULT_FLAGS_{n} is not in {MAKEFILE}")
    return m.group(1).split()


def part_flags(n, part):
    """flags that select one of the two objects: part 1 = external + dispatch, part 2 = in-LDS kernels (with their own flags)"""
    return [f"-DSMFFT_INST_PART={part}"] + (mult_flags(n) if part == 2 else [])


if __name__ == "__main__":
    print(" ".join(mult_flags(sys.argv[1] if len(sys.argv) > 1 else 1024)))
