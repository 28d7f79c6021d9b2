/*
 * ref_io_harness.cpp -- driver glue around the REAL reference's model I/O (test infrastructure).
 *
 * Compiled only where /root/reference exists (this container), by oracle/Makefile `refio`, against the reference's own
 * Optimizer/util_VSFM.h + DataInterface.h where they lie (g++ -fpermissive -include cstring), into oracle/_ref/ref_io.
 * Nothing here restates the loaders or writers: every parsed number and every output file comes out of reference code.
 *
 *   ref_io load   <in>  <dump>            LoadModelFile (by extension: .nvm / .out / BAL text) -> text dump of the parsed arrays
 *   ref_io loadid <in.nvm> <dump>         the id-passthrough overload LoadNVM(..., siftID, alldata)   util_VSFM.h:51-135
 *   ref_io save   <in>  <out> <fmt>       LoadModelFile, then fmt nvm  = SaveNVM, 7 arguments (feature id written as 0) :218-258
 *                                                       fmt nvmid = SaveNVM with Point3D measurements (id passthrough)   :261-308
 *                                                       fmt bal  = SaveBundlerModel :503-529, fmt out = SaveBundlerOut :407-446
 *   ref_io rip    <in>  <dump>            LoadModelFile, RemoveInvisiblePoints :736-866, dump (first line: its return value)
 * Dump format: one array per line, "name count v0 v1 ...", floats with %.9g (exact for float), ints as ints.
 */
#include "util_VSFM.h"
#include <cstdio>

static void dump(const char *path, int rc, vector<CameraT> &cam, vector<Point3D> &pt, vector<Point2D> &ms, vector<int> &ptidx,
                 vector<int> &camidx, vector<string> &names, vector<int> &ptc, vector<int> *sift)
{
    FILE *f = fopen(path, "w");
    fprintf(f, "rc 1 %d\n", rc);
    fprintf(f, "names %zu", names.size()); for (auto &n : names) fprintf(f, " %s", n.c_str()); fprintf(f, "\n");
    fprintf(f, "f %zu", cam.size()); for (auto &c : cam) fprintf(f, " %.9g", c.f); fprintf(f, "\n");
    fprintf(f, "R %zu", cam.size() * 9); for (auto &c : cam) for (int j = 0; j < 9; ++j) fprintf(f, " %.9g", c.m[j / 3][j % 3]); fprintf(f, "\n");
    fprintf(f, "t %zu", cam.size() * 3); for (auto &c : cam) for (int j = 0; j < 3; ++j) fprintf(f, " %.9g", c.t[j]); fprintf(f, "\n");
    fprintf(f, "radial %zu", cam.size()); for (auto &c : cam) fprintf(f, " %.9g", c.radial); fprintf(f, "\n");
    fprintf(f, "dtype %zu", cam.size()); for (auto &c : cam) fprintf(f, " %d", c.distortion_type); fprintf(f, "\n");
    fprintf(f, "xyz %zu", pt.size() * 3); for (auto &p : pt) for (int j = 0; j < 3; ++j) fprintf(f, " %.9g", p.xyz[j]); fprintf(f, "\n");
    fprintf(f, "rgb %zu", ptc.size()); for (int v : ptc) fprintf(f, " %d", v); fprintf(f, "\n");
    fprintf(f, "xy %zu", ms.size() * 2); for (auto &m : ms) fprintf(f, " %.9g %.9g", m.x, m.y); fprintf(f, "\n");
    fprintf(f, "pt %zu", ptidx.size()); for (int v : ptidx) fprintf(f, " %d", v); fprintf(f, "\n");
    fprintf(f, "cam %zu", camidx.size()); for (int v : camidx) fprintf(f, " %d", v); fprintf(f, "\n");
    if (sift) { fprintf(f, "fid %zu", sift->size()); for (int v : *sift) fprintf(f, " %d", v); fprintf(f, "\n"); }
    fclose(f);
}

int main(int argc, char **argv)
{
    if (argc < 4) return 2;
    const string cmd = argv[1];
    vector<CameraT> cam; vector<Point3D> pt; vector<Point2D> ms; vector<int> ptidx, camidx, ptc; vector<string> names;
    if (cmd == "loadid") {
        ifstream in(argv[2]);
        vector<int> sift; vector<vector<pair<pair<int, int>, pair<float, float>>>> all;
        const bool ok = LoadNVM(in, cam, pt, ms, ptidx, camidx, names, ptc, sift, all);
        dump(argv[3], ok, cam, pt, ms, ptidx, camidx, names, ptc, &sift);
        return 0;
    }
    const bool ok = LoadModelFile(argv[2], cam, pt, ms, ptidx, camidx, names, ptc);
    if (cmd == "load") { dump(argv[3], ok, cam, pt, ms, ptidx, camidx, names, ptc, nullptr); return 0; }
    if (cmd == "rip") {
        const bool r = RemoveInvisiblePoints(cam, pt, ptidx, camidx, ms, names, ptc);
        dump(argv[3], r, cam, pt, ms, ptidx, camidx, names, ptc, nullptr);
        return 0;
    }
    if (cmd == "save" && argc >= 5 && ok) {
        const string fmt = argv[4];
        if (fmt == "nvm") SaveNVM(argv[3], cam, pt, ms, ptidx, camidx, names, ptc);
        else if (fmt == "nvmid") {   // what the callers read back: "camIdx pointId x y" (MergeClusters.cpp:503-506); ids from the id overload
            vector<CameraT> c2; vector<Point3D> p2; vector<Point2D> m2; vector<int> pi2, ci2, pc2, sift; vector<string> n2;
            vector<vector<pair<pair<int, int>, pair<float, float>>>> all;
            ifstream in(argv[2]);
            LoadNVM(in, c2, p2, m2, pi2, ci2, n2, pc2, sift, all);
            vector<Point3D> m3(m2.size());
            for (size_t k = 0; k < m2.size(); ++k) m3[k].SetPoint((float)sift[k], m2[k].x, m2[k].y);
            SaveNVM(argv[3], c2, p2, m3, pi2, ci2, n2, pc2);
        } else if (fmt == "bal") SaveBundlerModel(argv[3], cam, pt, ms, ptidx, camidx);
        else if (fmt == "out") SaveBundlerOut(argv[3], cam, pt, ms, ptidx, camidx, names, ptc);
        else return 2;
        return 0;
    }
    return ok ? 2 : 1;
}
