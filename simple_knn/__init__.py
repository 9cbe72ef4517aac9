"""Drop-in for the reference's `simple_knn` package (src/gaussiansplatting/submodules/simple-knn): `simple_knn._C.distCUDA2`."""
