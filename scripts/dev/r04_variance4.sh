# physically contiguous workspace slab (GOLAP_SLAB_CONTIGUOUS=1) against plain hipMalloc, alternating, one box
for i in 1 2 3 4; do
  echo "contiguous"; GOLAP_DEBUG_ENV=1 GOLAP_SLAB_CONTIGUOUS=1 python scripts/dev/variance2.py 2 2>/dev/null || exit 1
  echo "plain"; python scripts/dev/variance2.py 2 2>/dev/null || exit 1
done
