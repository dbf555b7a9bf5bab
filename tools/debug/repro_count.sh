for t in "$@"; do
 echo "== $t"
 python - "$t" <<'PY' 2>&1 | grep -c "False$"
import sys, runpy
sys.path.insert(0, '.')
from spatiotemporalentropymodel_amd import _lib
for kv in sys.argv[1].split(','):
    k, v = kv.split('=')
    assert _lib.hip().stem_tuning_set(k.encode(), int(v)) == 0
sys.argv = ["repro_fwd.py", "30"]
runpy.run_path('tools/debug/repro_fwd.py', run_name='__main__')
PY
done
