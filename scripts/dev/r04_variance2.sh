for i in 1 2 3 4; do python scripts/dev/variance2.py 2 2>/dev/null || exit 1; done
echo "-- with a 200 GiB block taken and given back first"
for i in 1 2 3 4; do PREALLOC_GB=200 python scripts/dev/variance2.py 1 2>/dev/null || exit 1; done
