"""dev: per-step reach-set build of a lone problem and small batches: three-wave blocks against four-wave blocks
(ARMOUR_P1_WAVES), with and without the w_aux recursion on the fourth wave (ARMOUR_P1_AUX3); one child process per setting,
build ms best of 5."""
import os, subprocess, sys
sys.path.insert(0, '/root/repo')
if len(sys.argv) > 1 and sys.argv[1] == "child":
    from armour_amd.planner import ArmourNLP
    from armour_amd.worlds import random_batch
    for B in (3, 5, 6, 7, 8, 10, 11, 12, 14, 15, 16, 18, 20, 22):
        bp = random_batch(5, B, 20)
        nlp = ArmourNLP(T=100)
        ms = []
        for _ in range(5):
            nlp.set_parameters(bp["q0"], bp["qd0"], bp["qdd0"], bp["q_des"], bp["obstacles"]); ms.append(nlp.build_ms)
        print(f"  B={B}: {min(ms):.3f} ms", flush=True); nlp.close()
    sys.exit(0)
for env in ({}, {"ARMOUR_P1_WAVES": "1"}, {"ARMOUR_P1_WAVES": "4"}):
    e = dict(os.environ); e.update(env); e.setdefault("ARMOUR_P1_TV", "0")
    r = subprocess.run([sys.executable, __file__, "child"], env=e, capture_output=True, text=True, timeout=600)
    print(f"{env or 'default'}\n{r.stdout.rstrip()}\n{r.stderr.strip()[-300:] if r.returncode else ''}", flush=True)
