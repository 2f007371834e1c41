"""Drop-in for the third-party `simple_knn` extension the reference imports at scene/gaussian_model.py:20
(`from simple_knn._C import distCUDA2`; cloned at install time, reference README.md:24).  Backed by the
MI355X-native kernel in scgaussian_amd/csrc/knn.hip (include/scg_knn.h)."""
