for rep in 1 2 3; do for csr in 0 1; do echo "CSR $csr"; timeout -k 10 100 python tools/perf_probe.py --configs 512:1:512 --packed 1 --steps 3000 --csr $csr || exit 1; done; done
for csr in 0 1; do echo "CSR $csr"; timeout -k 10 100 python tools/perf_probe.py --configs 1:1:1024 --packed 0 --steps 3000 --csr $csr || exit 1; done
