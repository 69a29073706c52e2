"""Global mass balance from the device ledger.

Mirrors ``_mass_bal_global`` (/root/reference/src/clearwater_riverine/postproc_util.py:21-166): same quantities, same
column names, but the constituent-dependent sums (mass in the domain, mass through every boundary-condition line)
are reduced on the GPU step by step (cwr_domain_mass, CWR_STEP_MASS_BALANCE) instead of from RAM-resident (T, ncell) /
(T, nedge) histories.  The volume columns depend on the flow field only and are summed on the host.
"""
from __future__ import annotations

import numpy as np


def boundary_lines(boundary_data) -> list:
    """[(name, face ids)] sorted by 'BC Line ID' (postproc_util.py:72-90).  boundary_data: a pandas DataFrame with the
    reference's columns 'Name', 'BC Line ID', 'Face Index' (io/hdf.py:355-436), or a dict name -> face ids."""
    if isinstance(boundary_data, dict):
        return [(str(k), np.asarray(v, dtype=np.int64).ravel()) for k, v in boundary_data.items()]
    ids = boundary_data.groupby(by='Name').mean(numeric_only=True).sort_values(by=['BC Line ID']).reset_index()
    out = []
    for _, row in ids.iterrows():
        sel = boundary_data.loc[boundary_data['BC Line ID'] == row['BC Line ID']]
        out.append((str(row['Name']), sel['Face Index'].to_numpy().astype(np.int64).ravel()))
    return out


def volume_columns(face_flow, dt, lines) -> dict:
    """Per line: total / in (<= 0) / out (>= 0) volume = sum_t sum_faces face_flow * dt, NaN (the trailing dt)
    skipped as xarray's sum does (postproc_util.py:92-98,112-130)."""
    cols = {}
    for name, faces in lines:
        vol = np.asarray(face_flow)[:, faces] * np.asarray(dt, dtype=np.float64)[:, None]
        cols[name] = (np.nansum(vol), np.nansum(np.where(vol <= 0, vol, 0.0)), np.nansum(np.where(vol >= 0, vol, 0.0)))
    return cols


def assemble(lines, vol_cols, ledger_k, vol_start, mass_start, vol_end, mass_end) -> dict:
    """The row ``_mass_bal_global`` returns, as a dict with the reference's column names (postproc_util.py:60-165).
    ledger_k: (n_lines, 3) for ONE constituent."""
    d = {'Vol_start': vol_start, 'Mass_start': mass_start, 'Vol_end': vol_end, 'Mass_end': mass_end}
    tv = tvi = tvo = tm = tmi = tmo = 0.0
    for li, (name, _) in enumerate(lines):
        v, vi, vo = vol_cols[name]
        m, mi, mo = (float(x) for x in ledger_k[li])
        d[f'{name}_vol'], d[f'{name}_mass'] = v, m
        d[f'{name}_in_vol'], d[f'{name}_in_mass'] = vi, mi
        d[f'{name}_out_vol'], d[f'{name}_out_mass'] = vo, mo
        tv, tvi, tvo, tm, tmi, tmo = tv + v, tvi + vi, tvo + vo, tm + m, tmi + mi, tmo + mo
    d.update(bcTotalVolInOutAll=tv, bcTotalVolInAll=tvi, bcTotalVolOutAll=tvo,
             bcTotalMassInOutAll=tm, bcTotalMassInAll=tmi, bcTotalMassOutAll=tmo)
    with np.errstate(divide='ignore', invalid='ignore'):
        d['vol_end_calc'] = vol_start - tvi - tvo
        d['error_vol'] = d['vol_end_calc'] - vol_end
        d['prct_error_vol'] = np.float64(d['error_vol']) / np.float64(tvi) * 100
        d['mass_end_calc'] = mass_start - tmi - tmo
        d['error_mass'] = d['mass_end_calc'] - mass_end
        d['prct_error_mass'] = np.float64(d['error_mass']) / np.float64(tmi) * 100
    return d
