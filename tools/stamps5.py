import os, sys
sys.path.insert(0, "tools"); sys.path.insert(0, ".")
os.environ["GRIM_QUIET"]="1"
import harness, wmda_scale
wmda_scale.ensure()
name = wmda_scale.name_of()
n = int(sys.argv[1])
lines = wmda_scale.subjects(n)
for rep in range(2):
    got, log, imp = harness.run_product(name, wmda_scale.conf(), lines, tag="w5", on_unsupported="skip", quiet=True)
print(imp.last_stats)
