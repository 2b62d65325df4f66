# band height / prefetch depth sweep of the 9x9 sliding kernel (TRK_BLUR_RPB forces the band height, TRK_BLUR_D9 the depth-9 form)
for N in ${@:-2560 3072 3584 4096 4160 5120}; do
  echo "N=$N default: $(python tools/blur_sizes.py $N | cut -c9-32)"
  for r in 28 46 64 82 100; do echo "  D6 rpb=$r: $(TRK_BLUR_RPB=$r python tools/blur_sizes.py $N | cut -c14-24)"; done
  for r in 19 28 37 46 55 64 73 82; do echo "  D9 rpb=$r: $(TRK_BLUR_D9=1 TRK_BLUR_RPB=$r python tools/blur_sizes.py $N | cut -c14-24)"; done
done
