"""Worker of tests/test_gpu_00_world2.py::test_environment_switches_do_not_change_results: the library reads its environment
switches once per process, so each setting runs in a fresh process.  Clusters one IVF-regime dataset (flat and indexed buckets,
n_probe 16 and 5) and writes the labels / medoids to <out>.npz.  Never imported by pytest."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(out):
    from falcon_amd import synth
    from falcon_amd.cluster.cluster import AnnParams, ClusterPipeline, SpectrumDataset
    from falcon_amd.device import Context
    pipe = ClusterPipeline(Context(0))
    res = {}
    d = synth.select_charge(synth.generate(40000, seed=23, mz_lo=600.0, mz_hi=605.0), 2)      # ~5,000-row windows: n_list 128
    ds = SpectrumDataset(d["precursor_mz"], d["retention_time"], d["mz"], d["intensity"], d["indptr"])
    for np_ in (16, 5):
        lab, med = pipe.run(ds, 20.0, "ppm", None, 0.05, 2 ** 15, AnnParams(n_probe=np_))
        res[f"labels{np_}"], res[f"medoids{np_}"] = lab.cpu().numpy(), med.cpu().numpy()
        res[f"n_list_max{np_}"] = int(pipe.last["n_list"].max())
    np.savez(out, **res)


if __name__ == "__main__":
    main(sys.argv[1])
