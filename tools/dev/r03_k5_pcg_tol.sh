# K5 at C2: where PCG (symmetric operator) hands over to BiCGStab (exact operator) -- SSRS_SOLVE_PCG_TOL
cd $GRAFT_REPO_ROOT
for t in ${TOLS:-1 1e-2 1e-3 1e-4 1e-5 1e-6 1e-7 1e-8 1e-10 1e-15}; do
  echo "== PCG_TOL $t"; SSRS_SOLVE_PCG_TOL=$t python tools/dev/probe_k5_omegas.py "0.7,0.7" 2>&1 | grep "omegas"
done
