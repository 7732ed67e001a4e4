"""Rebuild a boundary fixture's case through the host mirror: TEST INFRASTRUCTURE."""
import json

import hfx_host as H


def groups_of(meta):
    """(bcs, sides) for hfx_host.Case from a fixture's meta: boundary groups in the mesh file's order
    (first appearance over the element-local face numbers, as oracle/gen_neu_mesh.py writes them)."""
    kk = meta["keys"]
    sides_names = H.SIDES3 if meta["dims"] == 3 else H.SIDES2
    names = []
    for side in sides_names:
        g = (meta.get("bcs") or {}).get(side, "Cyclic")
        if g not in names:
            names.append(g)
    bcs = []
    for g in names:
        b = {"type": kk["bc_%s_type" % g]}
        pre = "bc_%s_" % g
        for key, v in kk.items():
            if key.startswith(pre) and key != pre + "type":
                b[key[len(pre):]] = v
        bcs.append(b)
    sides = {s: names.index((meta.get("bcs") or {}).get(s, "Cyclic")) for s in sides_names}
    return bcs, sides


def case_from_fixture(d, **over):
    meta = json.loads(bytes(d["meta_json"]).decode())
    kk = meta["keys"]
    bcs, sides = groups_of(meta)
    n = meta["n"]
    n = [n] * meta["dims"] if isinstance(n, int) else n
    kw = dict(dims=meta["dims"], order=kk["order"], adv_type=kk["adv_type"], riemann_solve_type=kk["riemann_solve_type"],
              viscous=kk["viscous"], ic_form=kk["ic_form"], fix_vis=kk["fix_vis"], T_c_ic=kk["T_c_ic"], rho_c_ic=kk["rho_c_ic"],
              upts_type=kk["upts_type_hexa"], vcjh_scheme=kk["vcjh_scheme_hexa"])
    for k in ("u_c_ic", "v_c_ic", "w_c_ic", "p_c_ic", "over_int", "over_int_order", "shock_cap", "shock_det_field", "s0",
              "expf_fac", "expf_order", "expf_cutoff", "dt"):
        if k in kk:
            kw[k] = kk[k]
    if meta["dims"] == 2:
        kw["upts_type"], kw["vcjh_scheme"] = kk["upts_type_quad"], kk["vcjh_scheme_quad"]
    kw.update(over)
    return H.Case(n + [1] * (3 - len(n)), xv=d["xv"], bcs=bcs, sides=sides, **kw), meta
