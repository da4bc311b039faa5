import ctypes, os
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libhandoff.so"))
res = (ctypes.c_ulonglong * 4)()
names_st = {0: "store sc1(agent)", 1: "store plain"}
names_ld = {0: "load sc1(agent)", 1: "load nt", 2: "load plain"}
for same in (1, 0):
    for st in (0, 1):
        for ld in (0, 1, 2):
            lib.run_pingpong(st, ld, same, 2000, res)
            print(f"{'same XCD' if same else 'diff XCD'} | {names_st[st]:18s} | {names_ld[ld]:16s} | round trip {res[0]:8d} cycles  bad={res[1]} xcc0={res[2]} partner_block={res[3]}")
