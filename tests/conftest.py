import hashlib
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built():
    """Everything compiled (HIP lib cross-compiles on CPU; the oracle and the emulation are host code)."""
    import __graft_entry__ as G
    G.build()
    return True


@pytest.fixture(scope="session")
def demo(built):
    """data/demo: index built from the committed demo zips (tools/make_demo_index.sh) + the demo FASTQ."""
    import __graft_entry__ as G
    d = G.demo_dir()
    return {"dir": d, "index": os.path.join(d, "index"), "fastq": os.path.join(d, "ERR1050068.fastq")}


@pytest.fixture(scope="session")
def strain(built):
    """data/strain: a second, synthetic index (strains + tandem repeats, tools/make_strain_index.sh) and 96 ONT reads
    simulated from it; the expected SAM (reference, UB-pinned build) is tests/golden/strain/reads.ubfree.sam."""
    d = os.path.join(ROOT, "data", "strain")
    subprocess.check_call([os.path.join(ROOT, "tools", "make_strain_index.sh"), d], stdout=subprocess.DEVNULL)
    fq = os.path.join(d, "reads.fq")
    assert md5_file(fq) == open(os.path.join(GOLDEN, "strain", "reads.fq.md5")).read().strip(), "strain read set differs from the one the golden SAM was made from"
    return {"index": os.path.join(d, "index"), "fastq": fq, "sam": os.path.join(GOLDEN, "strain", "reads.ubfree.sam")}


@pytest.fixture(scope="session")
def oracle(demo):
    import oracle_lib
    return oracle_lib.Oracle(demo["index"])


@pytest.fixture(scope="session")
def golden_md5():
    return open(os.path.join(GOLDEN, "demo_sam.md5")).read().strip()


def md5_file(path):
    return hashlib.md5(open(path, "rb").read()).hexdigest()


def sam_lines(path):
    return open(path, "rb").read().splitlines()
