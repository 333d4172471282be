"""rocprofv3 kernel_trace.csv -> compact timeline (queue, stream, start_us, dur_us, short kernel name), sorted by start."""
import csv, sys, re
src, dst = sys.argv[1], sys.argv[2]
rows = []
with open(src) as f:
    for r in csv.DictReader(f):
        name = r["Kernel_Name"]
        name = re.sub(r"^void ", "", name)
        name = name.replace("(bool)", "").replace("(int)", "").replace("Cfg", "C")
        name = re.sub(r"\([A-Za-z_][^()]*\)$", "", name)[:110].replace(",", ";")
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), r.get("Stream_Id", "?"), name))
rows.sort()
t0 = rows[0][0]
with open(dst, "w") as f:
    f.write("queue,stream,start_us,dur_us,name\n")
    for s, e, q, st, n in rows:
        f.write(f"{q},{st},{(s - t0) / 1e3:.2f},{(e - s) / 1e3:.2f},{n}\n")
print(len(rows), "kernels")
