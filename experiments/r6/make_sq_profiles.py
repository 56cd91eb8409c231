"""gpurun_out/r6_sq/<config>_bilinear_<suffix>_p{1,2,3}.txt (experiments/r6/sq_all.sh <suffix>) -> profiles/r06_<config>_bilinear_sq.txt: the digest
(experiments/r5/sq_digest.py) over the three raw passes.   python experiments/r6/make_sq_profiles.py <suffix>"""
import subprocess, sys, os
suffix = sys.argv[1] if len(sys.argv) > 1 else "final"
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
src = os.path.join(root, "gpurun_out", "r6_sq")
for c in ("c1", "c2", "c3", "c5"):
    tag = f"{c}_bilinear_{suffix}"
    if not os.path.exists(os.path.join(src, tag + "_p1.txt")):
        continue
    digest = subprocess.run([sys.executable, os.path.join(root, "experiments", "r5", "sq_digest.py"), src, tag, "bilinear"], capture_output=True, text=True, check=True).stdout
    with open(os.path.join(root, "profiles", f"r06_{c}_bilinear_sq.txt"), "w") as out:
        out.write("# SQ counters, MI355X, round 6 FINAL kernels (pair layout of the stitch, integer blend, workgroups of two waves where the plan's windows allow): three\n"
                  "# rocprofv3 --pmc passes (kernel-trace only, the program directly after --) taken by experiments/r6/sq_all.sh " + suffix + "; digest by experiments/r5/sq_digest.py;\n"
                  f"# round 5's figures: profiles/r05_{c}_bilinear_final_sq.txt\n")
        out.write(digest + "\n# raw\n")
        for i in (1, 2, 3):
            out.write(f"## pass {i}\n" + open(os.path.join(src, f"{tag}_p{i}.txt")).read())
    print("wrote", f"profiles/r06_{c}_bilinear_sq.txt")
