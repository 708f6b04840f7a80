for nd in 200 120; do for w in 4 5 6 7 8; do
  MCMC_WAVE_WPB=$w timeout -k 10 240 python bench.py --n-data $nd --chains-per-gpu 8192 --mapping wave --no-cpu-baseline --ess-chains 0 --no-fast-leg --no-strict-leg --no-toy-ess 2>/dev/null | python tools/bench_line.py "n_data=$nd chains=8192 wpb=$w" || exit 1
done; done
