# under the condition that was slow 4 times out of 4 (a 200 GiB block taken and given back first): does the stride between the first
# pass's output sub-ranges decide it?
for pad in 0 2048 8192 32768 0 8192; do
  echo "GOLAP_CAP_PAD=$pad"
  PREALLOC_GB=200 GOLAP_DEBUG_ENV=1 GOLAP_CAP_PAD=$pad python scripts/dev/variance2.py 1 2>/dev/null || exit 1
done
