"""The C-ABI library loads without a GPU and exports every symbol include/hfx.h declares."""
import ctypes as C

import hfx


def test_library_exports_every_declared_symbol():
    names = hfx.declared_symbols()
    assert len(names) >= 25
    lib = hfx.lib()
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing


def test_struct_layouts_match_oracle_binding():
    # hfx_params and orc_params are declared field-for-field alike (include/hfx.h, oracle/oracle.h)
    import oracle_py as O
    assert C.sizeof(hfx.Params) == C.sizeof(O.Params)
    assert [f[0] for f in hfx.Params._fields_] == [f[0] for f in O.Params._fields_]


def test_version():
    assert hfx.lib().hfx_version() >= 1
