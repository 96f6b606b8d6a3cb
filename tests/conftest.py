import gzip
import os
import shutil
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLDEN = os.path.join(HERE, "golden")
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box via gpurun)")


def _unpack_db(tag, dst):
    for ext in ("bas", "seq", "acc", "nam", "ind"):
        with gzip.open(os.path.join(GOLDEN, f"{tag}db.{ext}.gz"), "rb") as f, open(os.path.join(dst, f"{tag}db.{ext}"), "wb") as g:
            shutil.copyfileobj(f, g)
    with gzip.open(os.path.join(GOLDEN, f"{tag}.stg.gz"), "rb") as f, open(os.path.join(dst, f"{tag}.stg"), "wb") as g:
        shutil.copyfileobj(f, g)


@pytest.fixture(scope="session")
def golden_dir(tmp_path_factory):
    """tests/golden with the gzip'd reference-built DBs and stage dumps unpacked."""
    d = tmp_path_factory.mktemp("golden")
    for tag in ("c1", "mix", "quirk"):
        _unpack_db(tag, str(d))
    return str(d)


@pytest.fixture(scope="session")
def oracle():
    import oraclelib
    oraclelib.build()
    oraclelib.lib()
    return oraclelib
