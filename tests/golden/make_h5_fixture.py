"""Convert the reference's HDF5 test world (test/data/june_world.h5) into a neutral .npz fixture.

h5py is not installed for the build's interpreter; the image's conda Python has it.  Run once in the
build container:

    /opt/conda/bin/python3.9 tests/golden/make_h5_fixture.py

Only the datasets the reference's june_world_loader reads are kept."""
import os

import h5py
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = "/root/reference/test/data/june_world.h5"
KEEP = {
    "population": ["id", "age", "sex", "ethnicity", "area", "super_area", "group_ids", "group_specs"],
    "households": ["id"], "care_homes": ["id"], "companies": ["id"], "schools": ["id"], "universities": ["id"],
    "geography": ["super_area_coordinates", "super_area_id", "area_name", "area_socioeconomic_indices"],
}

out = {}
with h5py.File(SRC, "r") as f:
    for group, names in KEEP.items():
        for name in names:
            a = f[group][name][:]
            if a.dtype.kind in "SO":
                a = a.astype("U")
            out[f"{group}/{name}"] = a
np.savez_compressed(os.path.join(HERE, "june_world_h5.npz"), **out)
print("wrote june_world_h5.npz with", len(out), "datasets")
