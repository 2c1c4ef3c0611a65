"""Developer options of the C ABI (include/graphminer_amd.h gm_dev_option / gm_dev_option_get): the library reads no algorithm switch from the
environment -- what tests and A/B runs need is set by name through the ABI.  CPU only: set / get / replace / remove / clear, and the
environment is not a fallback in the shipped build."""
import os

from graphminer_amd import _lib


def test_set_get_replace_remove_clear(monkeypatch):
    lib = _lib.load()
    _lib.dev_option(None)
    assert lib.gm_dev_option_get(b"GM_SUP_MASK_MIN") is None
    _lib.dev_option("GM_SUP_MASK_MIN", 7)
    _lib.dev_option("GM_TOPO_MIN_ROW", "0")
    assert lib.gm_dev_option_get(b"GM_SUP_MASK_MIN") == b"7" and lib.gm_dev_option_get(b"GM_TOPO_MIN_ROW") == b"0"
    _lib.dev_option("GM_SUP_MASK_MIN", 192)
    assert lib.gm_dev_option_get(b"GM_SUP_MASK_MIN") == b"192"
    _lib.dev_option("GM_SUP_MASK_MIN", None)
    assert lib.gm_dev_option_get(b"GM_SUP_MASK_MIN") is None and lib.gm_dev_option_get(b"GM_TOPO_MIN_ROW") == b"0"
    _lib.dev_option(None)
    assert lib.gm_dev_option_get(b"GM_TOPO_MIN_ROW") is None
    assert lib.gm_dev_option(b"", b"1") == _lib.GM_ERR_INVALID
    # the environment is no switch: a variable of an option's name changes nothing (make DEVEL=1 builds fall back to it; this one is not)
    monkeypatch.setenv("GM_FORCE_RCCL_PATH", "1")
    assert lib.gm_dev_option_get(b"GM_FORCE_RCCL_PATH") is None


def test_no_getenv_of_an_algorithm_switch_in_the_library_sources():
    """every getenv left in csrc/ is GM_SETUP_TRACE (diagnostics) or sits behind GM_DEVEL"""
    import re

    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "graphminer_amd")
    for sub in ("csrc", "host"):
        for f in sorted(os.listdir(os.path.join(root, sub))):
            src = open(os.path.join(root, sub, f)).read()
            for m in re.finditer(r"(?<![A-Za-z_])getenv\(([^)]*)\)", src):
                arg = m.group(1)
                devel = "#ifdef GM_DEVEL" in src[max(0, m.start() - 200):m.start()]
                assert "GM_SETUP_TRACE" in arg or devel, (f, m.group(0))
