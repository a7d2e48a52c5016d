for rep in 1 2; do for t in "" "ntt_tile_b_wide=0" "ntt_cols_per_wg=16" "ntt_cols_per_wg=4"; do
  python tools/ldebench.py --cols 256 --reps 20 --tunables "$t" --tag "$t" 2>/dev/null | grep expand
done; done
