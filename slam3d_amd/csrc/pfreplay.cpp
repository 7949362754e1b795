// pfreplay -- CSV replay driver over libparticlefilter.so, with file names, particle count, seed and constants given
// at run time.  It does what the reference's three replay programs do with everything hard-coded:
//   loc  : test/test.c:90-105 and csvlocalize/csvlocalize.c:64-88 (simple `t,y,z,x` / `t,beacon,range` rows, a deployment
//          file `beacon,y,z,x`; output rows `t,y,z,x,theta`, test.c:213)
//   slam : csvslam/csvslam.c:54-95 (verbose `t,position,waypoint,acc,y,z,x` line pairs and `t,uwb_range,waypoint,name,range`
//          rows, beacon = first letter of the name; output tag.csv rows `t,x,y,z,theta` and bcn.csv rows `b,x,y,z,theta`)
// Host code only: every particle operation goes through the C-ABI (include/particleFilter.h, include/pfx.h).
//
//   pfreplay loc  --vio F --uwb F --deploy F --out F [--format simple|verbose] [--n N] [--seed S] [--beacons 4]
//                 [--bias 0.2] [--std 0.1] [--range-only] [--device D]
//   pfreplay slam --vio F --uwb F --out-tag F --out-bcn F [--seed S] [--beacons 12] [--bias 0.4] [--std 0.1]
//                 [--no-skip] [--device D]
#include "../../include/pfx.h"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

namespace {

struct VioRow { double t; float x, y, z; char wp; };
struct UwbRow { double t; int b; float r; char wp; };

[[noreturn]] void fail(const std::string& msg)
{
    fprintf(stderr, "pfreplay: %s\n", msg.c_str());
    exit(2);
}

std::vector<std::string> split(const std::string& line)
{
    std::vector<std::string> out;
    size_t a = 0;
    while (true) {
        const size_t b = line.find(',', a);
        out.push_back(line.substr(a, b == std::string::npos ? std::string::npos : b - a));
        if (b == std::string::npos) break;
        a = b + 1;
    }
    return out;
}

std::vector<std::string> read_lines(const std::string& path)
{
    FILE* f = fopen(path.c_str(), "r");
    if (!f) fail("cannot open " + path);
    std::vector<std::string> lines;
    char buf[1024];                                 // LINE_LEN of the reference programs
    while (fgets(buf, sizeof buf, f)) {
        std::string s(buf);
        while (!s.empty() && (s.back() == '\n' || s.back() == '\r')) s.pop_back();
        lines.push_back(s);
    }
    fclose(f);
    return lines;
}

// test.c:164-176 (simple) / csvslam.c:97-119 (verbose: position line, then an orientation line that is skipped)
std::vector<VioRow> read_vio(const std::string& path, bool verbose)
{
    std::vector<VioRow> rows;
    const auto lines = read_lines(path);
    for (size_t k = 0; k < lines.size(); k += verbose ? 2 : 1) {
        const auto c = split(lines[k]);
        if (c.size() < (verbose ? 7u : 4u)) continue;
        const int o = verbose ? 4 : 1;              // columns y, z, x (iOS order)
        rows.push_back(VioRow{ atof(c[0].c_str()), (float)atof(c[o + 2].c_str()), (float)atof(c[o].c_str()),
                               (float)atof(c[o + 1].c_str()), verbose ? c[2][0] : '9' });
    }
    return rows;
}

// test.c:178-189 (simple) / csvslam.c:121-140 (verbose)
std::vector<UwbRow> read_uwb(const std::string& path, bool verbose, int numBcns)
{
    std::vector<UwbRow> rows;
    for (const auto& line : read_lines(path)) {
        const auto c = split(line);
        if (c.size() < (verbose ? 5u : 3u)) continue;
        UwbRow r;
        r.t = atof(c[0].c_str());
        r.b = verbose ? (c[3].empty() ? -1 : c[3][0] - 'a') : atoi(c[1].c_str());
        r.r = (float)atof(c[verbose ? 4 : 2].c_str());
        r.wp = verbose ? c[2][0] : '9';
        if (r.b < 0 || r.b >= numBcns) fail("beacon id out of range in " + path + ": " + line);     // the reference asserts
        rows.push_back(r);
    }
    return rows;
}

// csvslam.c:116,135: the first read skips rows whose waypoint field starts with a character below '4'
template <typename T> void skip_to_waypoint(std::vector<T>& rows)
{
    size_t k = 0;
    while (k < rows.size() && rows[k].wp < '4') ++k;
    rows.erase(rows.begin(), rows.begin() + (long)k);
}

struct Args {
    std::map<std::string, std::string> kv;
    bool has(const char* k) const { return kv.count(k) != 0; }
    std::string str(const char* k, const char* dflt = nullptr) const
    {
        auto it = kv.find(k);
        if (it != kv.end()) return it->second;
        if (!dflt) fail(std::string("missing --") + k);
        return dflt;
    }
    double num(const char* k, double dflt) const { return has(k) ? atof(kv.at(k).c_str()) : dflt; }
};

int run_loc(const Args& a)
{
    const bool verbose = a.str("format", "simple") == "verbose";
    const int numBcns = (int)a.num("beacons", 4);                  // test.c:24
    const float bias = (float)a.num("bias", 0.2), stdr = (float)a.num("std", 0.1);      // test.c:25-26
    const long long n = (long long)a.num("n", PF_N_TAG_LOC);
    const unsigned seed = (unsigned)a.num("seed", 123456789);      // test.c:36
    auto vio = read_vio(a.str("vio"), verbose);
    auto uwb = read_uwb(a.str("uwb"), verbose, numBcns);
    if (a.has("range-only")) vio.clear();
    std::vector<float> dep((size_t)numBcns * 3, 0.0f);
    {   // test.c:191-205: rows `beacon,y,z,x`
        const auto lines = read_lines(a.str("deploy"));
        for (int i = 0; i < numBcns && i < (int)lines.size(); ++i) {
            const auto c = split(lines[(size_t)i]);
            if (c.size() < 4) continue;
            const int b = atoi(c[0].c_str());
            if (b < 0 || b >= numBcns) fail("deployment: beacon id out of range");
            dep[(size_t)b * 3 + 1] = (float)atof(c[1].c_str());
            dep[(size_t)b * 3 + 2] = (float)atof(c[2].c_str());
            dep[(size_t)b * 3 + 0] = (float)atof(c[3].c_str());
        }
    }
    pfx_loc* pf = pfx_loc_create(n, seed, 0, (int)a.num("device", 0));
    if (!pf) fail(std::string("pfx_loc_create: ") + pfx_last_error());
    FILE* out = fopen(a.str("out").c_str(), "w");
    if (!out) fail("cannot write " + a.str("out"));
    bool header = false;
    size_t iv = 0, iu = 0, nRange = 0, nRows = 0;
    // --repeat K: the event lists are replayed K times (timing runs; the time stamps start over, which the filter
    // accepts: only VIO differences enter it).  --warm W: W untimed passes first, so the loop time excludes the
    // first launches of the process (code object load).
    const int repeat = (int)a.num("repeat", 1), warm = (int)a.num("warm", 0);
    auto t0 = std::chrono::steady_clock::now();
    for (int rep = -warm; rep < repeat; ++rep) {
    if (rep == 0 && warm > 0) {
        if (pfx_loc_sync(pf)) fail(std::string("pfx_loc_sync: ") + pfx_last_error());
        nRange = 0; nRows = 0;
        t0 = std::chrono::steady_clock::now();
    }
    iv = 0; iu = 0;
    while (iv < vio.size() || iu < uwb.size()) {
        if (iv < vio.size() && (iu >= uwb.size() || vio[iv].t < uwb[iu].t)) {
            const VioRow& v = vio[iv++];
            pfx_loc_deposit_vio(pf, v.t, v.x, v.y, v.z, 0.0f);
            double t; float x, y, z, th;
            const int rc = pfx_loc_get_tag_loc(pf, &t, &x, &y, &z, &th);
            if (rc < 0) fail(std::string("pfx_loc_get_tag_loc: ") + pfx_last_error());
            if (rc) {
                if (!header) { fprintf(out, "t,x,y,z,theta\n"); header = true; }
                fprintf(out, "%lf,%f,%f,%f,%f\n", t, y, z, x, th);          // test.c:213
                ++nRows;
            }
        } else {
            const UwbRow& u = uwb[iu++];
            const float r = u.r - bias;
            if (r > 0.0f && r < 30.0f) {
                if (pfx_loc_deposit_range(pf, dep[(size_t)u.b * 3], dep[(size_t)u.b * 3 + 1], dep[(size_t)u.b * 3 + 2], r, stdr))
                    fail(std::string("pfx_loc_deposit_range: ") + pfx_last_error());
                ++nRange;
            }
        }
    }
    }
    double t; float x, y, z, th;
    const int ok = pfx_loc_get_tag_loc(pf, &t, &x, &y, &z, &th);
    const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    fclose(out);
    printf("pfreplay loc: N=%lld, %zu VIO rows, %zu ranges, %zu output rows, loop took %f seconds\n", n, iv, nRange, nRows, secs);
    if (repeat > 1 || warm > 0)
        printf("timing: %d timed passes, %.3f us per event (%zu range deposits + %zu pose reads)\n", repeat,
               1e6 * secs / (double)std::max<size_t>(1, nRange + nRows), nRange, nRows);
    if (ok == 1) printf("final pose: x=%f y=%f z=%f theta=%f\n", x, y, z, th);
    pfx_loc_destroy(pf);
    return 0;
}

int run_slam(const Args& a)
{
    const int numBcns = (int)a.num("beacons", 12);                 // csvslam.c:18
    const float bias = (float)a.num("bias", 0.4), stdr = (float)a.num("std", 0.1);      // csvslam.c:19-20
    if (numBcns < 1 || numBcns > 32) fail("--beacons must be in [1, 32]");
    auto vio = read_vio(a.str("vio"), true);
    auto uwb = read_uwb(a.str("uwb"), true, numBcns);
    if (!a.has("no-skip")) { skip_to_waypoint(vio); skip_to_waypoint(uwb); }   // SKIP_TO_WAYPOINT, csvslam.c:21
    if (a.has("seed")) particleFilterSeed_set((unsigned)a.num("seed", 0));
    setenv("PF_HIP_DEVICE", a.str("device", "0").c_str(), 1);
    // the reference-layout structs, owned by the caller as in csvslam.c:34-36
    particleFilterSlam_t* pf = (particleFilterSlam_t*)calloc(1, sizeof(particleFilterSlam_t));
    bcn_t* bcns = (bcn_t*)calloc((size_t)numBcns, sizeof(bcn_t));
    std::vector<bcn_t*> ptrs((size_t)numBcns);
    if (!pf || !bcns) fail("out of memory");
    particleFilterSlam_init(pf);
    for (int i = 0; i < numBcns; ++i) { particleFilterSlam_addBcn(&bcns[i]); ptrs[(size_t)i] = &bcns[i]; }
    FILE* tagOut = fopen(a.str("out-tag").c_str(), "w");
    FILE* bcnOut = fopen(a.str("out-bcn").c_str(), "w");
    if (!tagOut || !bcnOut) fail("cannot write the output files");
    bool header = false;
    size_t iv = 0, iu = 0, nRange = 0;
    const auto t0 = std::chrono::steady_clock::now();
    double t; float x, y, z, th;
    while (iv < vio.size() || iu < uwb.size()) {
        if (iv < vio.size() && (iu >= uwb.size() || vio[iv].t < uwb[iu].t)) {
            const VioRow& v = vio[iv++];
            particleFilterSlam_depositTagVio(pf, v.t, v.x, v.y, v.z, 0.0f);
            if (particleFilterSlam_getTagLoc(pf, &t, &x, &y, &z, &th)) {
                if (!header) { fprintf(tagOut, "t,x,y,z,theta\n"); header = true; }
                fprintf(tagOut, "%lf,%f,%f,%f,%f\n", t, x, y, z, th);       // csvslam.c:150
            }
        } else {
            const UwbRow& u = uwb[iu++];
            const float r = u.r - bias;
            if (r > 0.0f && r < 30.0f) {
                particleFilterSlam_depositRange(pf, &bcns[u.b], r, stdr, ptrs.data(), numBcns);
                ++nRange;
            }
        }
    }
    header = false;
    for (int b = 0; b < numBcns; ++b) {
        if (particleFilterSlam_getBcnLoc(pf, &bcns[b], &t, &x, &y, &z, &th)) {
            if (!header) { fprintf(bcnOut, "b,x,y,z,theta\n"); header = true; }
            fprintf(bcnOut, "%hhu,%f,%f,%f,%f\n", (unsigned char)b, x, y, z, th);   // csvslam.c:158
        }
    }
    const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    fclose(tagOut);
    fclose(bcnOut);
    printf("pfreplay slam: %d beacons, %zu VIO rows, %zu ranges, loop took %f seconds\n", numBcns, iv, nRange, secs);
    pfx_slam_release(pf);
    free(bcns);
    free(pf);
    return 0;
}

}  // namespace

int main(int argc, char** argv)
{
    if (argc < 2 || (strcmp(argv[1], "loc") && strcmp(argv[1], "slam"))) {
        fprintf(stderr, "usage: pfreplay loc|slam --vio F --uwb F ... (see the header of slam3d_amd/csrc/pfreplay.cpp)\n");
        return 2;
    }
    Args a;
    for (int k = 2; k < argc; ++k) {
        if (strncmp(argv[k], "--", 2)) fail(std::string("unexpected argument ") + argv[k]);
        const std::string key(argv[k] + 2);
        if (key == "range-only" || key == "no-skip") { a.kv[key] = "1"; continue; }
        if (k + 1 >= argc) fail("--" + key + " needs a value");
        a.kv[key] = argv[++k];
    }
    return strcmp(argv[1], "loc") ? run_slam(a) : run_loc(a);
}
