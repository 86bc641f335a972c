# per variant: the sweeps' stage table (tools/time_sweeps.py)
set -u
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
for so in build/var/libppk_v*.so; do
  echo "== $so"
  PPK_LIBRARY=$PWD/$so python tools/time_sweeps.py 2>&1 | grep "sort\|^1D\|^2D"
done
