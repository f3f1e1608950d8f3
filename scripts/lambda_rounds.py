import csv, sys, glob
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
lam0 = sorted([(int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for r in rows if "asm_lambda_tile_k<0>" in r["Kernel_Name"]])
lam1 = sorted([(int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for r in rows if "asm_lambda_tile_k<1>" in r["Kernel_Name"]])
g = sorted([(int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for r in rows if "gemm_nt_f64" in r["Kernel_Name"]])
n = len(lam0)
# the timed call is the last one: its rounds are the last block of launches; print the last 16 launches
print("lambda<0> us (last 14):", [round(x[1]) for x in lam0[-14:]])
print("lambda<1> us (last 14):", [round(x[1]) for x in lam1[-14:]])
print("gemm64 us (last 18):", [round(x[1]) for x in g[-18:]])
