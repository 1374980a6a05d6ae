# every fuzzer under scripts/ with the given seeds (default 11 12), one after the other on one box; prints each one's last line.
# bash scripts/fuzz_all.sh [seed ...]      (fuzz_denoise gets N=40, the others their defaults)
cd $GRAFT_REPO_ROOT
SEEDS=${@:-11 12}
rc=0
for s in $SEEDS; do
  for f in fuzz_demosaic fuzz_xtrans fuzz_sizes fuzz_tools fuzz_nlm fuzz_pixel fuzz_dninfo fuzz_batch_io; do
    out=$(SEED=$s timeout 600 python scripts/$f.py 2>&1 | tail -1); r=$?
    echo "[$f seed $s] $out"
  done
  out=$(SEED=$s N=40 timeout 900 python scripts/fuzz_denoise.py 2>&1 | tail -1)
  echo "[fuzz_denoise seed $s] $out"
done
