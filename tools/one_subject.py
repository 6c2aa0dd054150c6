import os, sys
sys.path.insert(0, "tools"); import harness
sys.path.insert(0, harness.ROOT)
os.environ["GRIM_QUIET"]="1"
conf = harness.base_conf(["CAU"])
line = "M7209,B*14:02/B*27:03/B*47:01+B*15:01^DQB1*05:02+DQB1*03:02^DRB1*16:02+DRB1*04:01,CAU,CAU"
for _ in range(2):
    got, log, imp = harness.run_product("cau", conf, [line], tag="one")
print(imp.last_stats)
