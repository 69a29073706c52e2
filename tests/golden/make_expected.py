#!/usr/bin/env python3
"""Expected OUTPUTS for the golden inputs, produced by the CPU oracle (oracle/cwr_oracle.py).

    python tests/golden/make_expected.py

The reference package itself cannot be imported in the build image (xarray, holoviews, geoviews,
geopandas are not installed; no network), so these are outputs of the restatement, pinned against
the reference's own fixture facts and notebook-printed known answers by tests/test_oracle.py.
Set-up = the reference's own test set-up (tests/test_riverine.py:89-127 of the reference): initial
condition from the IC CSV, boundary value 100 on the ghost cell of every boundary-line face.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(HERE)), 'oracle'))
sys.path.insert(0, os.path.dirname(HERE))
import cwr_oracle as oracle  # noqa: E402
from util import load_plan  # noqa: E402

CASES = {'plan01': (0.01, 30), 'plan02': (0.01, 24), 'plan03': (0.001, 30)}

if __name__ == '__main__':
    for plan, (D, steps) in CASES.items():
        mesh, inp, _ = load_plan(plan, D)
        model = oracle.OracleModel(mesh, {'c': inp})
        nnz = []
        for _ in range(steps):
            model.update()
            nnz.append(model.last_A.nnz)
        con = model.constituent_dict['c']
        out = os.path.join(HERE, f'{plan}_expected.npz')
        np.savez_compressed(out, diffusion_coefficient=D, steps=steps, state=con.state[:steps + 1],
                            advection_mass_flux=con.advection_mass_flux[:steps],
                            diffusion_mass_flux=con.diffusion_mass_flux[:steps],
                            total_mass_flux=con.total_mass_flux[:steps], nnz=np.array(nnz),
                            advection_coeff=mesh['advection_coeff'][:steps + 1],
                            coeff_to_diffusion=mesh['coeff_to_diffusion'][:steps + 1],
                            face_to_face_dist=mesh['face_to_face_dist'], dt=mesh['dt'])
        print(plan, 'steps', steps, '->', os.path.getsize(out), 'bytes')
