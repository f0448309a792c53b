# contiguous workspace (the reproducibly slow placement: first pass 5.0 instead of 3.2 ms per side): which stride between the first
# pass's output sub-ranges (GOLAP_CAP_PAD rows on top of the provisioned cap) avoids the channel conflicts?
for pad in 0 16 32 48 64 80 96 112 128 144 256 272 512 528 1024 1040 4096 4112; do
  echo "GOLAP_CAP_PAD=$pad"
  GOLAP_DEBUG_ENV=1 GOLAP_SLAB_CONTIGUOUS=1 GOLAP_CAP_PAD=$pad python scripts/dev/variance2.py 1 2>/dev/null | cut -c1-90 || exit 1
done
