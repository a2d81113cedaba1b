// comparator -- drop-in command line of the reference's comparator
// (Comparator.cpp:464-521) over libspsp: same flags, defaults, messages and
// output files (<o>_containment.csv.gz, <o>_jaccard.csv.gz).
#include <getopt.h>

#include <chrono>
#include <iostream>
#include <sstream>
#include <algorithm>
#include <cstdlib>
#include <string>
#include <vector>

#include "../../include/spsp.h"

using namespace std;

// Comparator::getfilesname (Comparator.cpp:7-21): lines longer than 2 chars
static bool read_names(const string& fof, vector<string>& out) {
    uint8_t* data = nullptr; uint64_t len = 0;
    if (spsp_read_file_host(fof.c_str(), &data, &len) != SPSP_OK) { cout << "Can't open " << fof << endl; return false; }
    istringstream is(string((const char*)data, len));
    spsp_free(data);
    string line;
    while (getline(is, line))
        if (line.size() > 2) out.push_back(line);
    return true;
}

int main(int argc, char** argv) {
    int ch;
    string inputfof, query, output_name("results");
    uint64_t p = 6;
    double min_threshold = 0;
    while ((ch = getopt(argc, argv, "hdag:q:k:m:n:s:t:b:e:f:i:p:o:")) != -1) {
        switch (ch) {
            case 'f': inputfof = optarg; break;
            case 'q': query = optarg; break;
            case 'p': p = stoi(optarg); break;
            case 'm': min_threshold = stod(optarg); break;
            case 'o': output_name = optarg; break;
        }
    }
    if (inputfof == "") {
        cout << "Core arguments:" << endl
             << "-f Index file of files (mandatory)" << endl
             << "-q Query file of files (\"\" for all versus all comparison of the index)" << endl
             << "Ouput arguments:" << endl
             << "-m Minimum value to be output (0.0)" << endl
             << "-p Required precision to be output in the CSV (6)" << endl
             << "-o output prefix (results)" << endl;
        return 0;
    }
    vector<string> names;
    uint32_t n_query = 0;
    if (query == "") {
        cout << "No query file, I will perform a all versus all comparison" << endl;
        read_names(inputfof, names);
        cout << "I found " << names.size() << " documents" << endl;
        n_query = (uint32_t)names.size();
    } else {
        read_names(query, names);
        n_query = (uint32_t)names.size();
        cout << "I query " << n_query << " file(s) against the bank" << endl;
        read_names(inputfof, names);
    }
    vector<const char*> paths;
    for (auto& s : names) paths.push_back(s.c_str());
    // Devices: every visible GPU when there is enough work to split (the comparison is dealt by key over one context per
    // device, spsp_compare_files_multi), else the first.  SPSP_DEVICES="0,1,2,3" (or "0,0": two contexts on one device)
    // names them explicitly.  The reference has no such notion (one thread, Comparator.cpp:39-74).
    vector<int> devices;
    if (const char* e = getenv("SPSP_DEVICES")) {
        istringstream is(e);
        string tok;
        while (getline(is, tok, ',')) if (!tok.empty()) devices.push_back(atoi(tok.c_str()));
    }
    if (devices.empty()) {
        const int visible = spsp_device_count();
        if (visible <= 0) { cout << "GPU unavailable: " << spsp_last_error() << endl; return 1; }
        // sketches per device below which splitting does not pay (one partition + one exchange + a host join per device against
        // a comparison of a few hundred microseconds); SPSP_PER_DEVICE=<n> overrides the 512 (a guess until measured on a node)
        size_t per_device = 512;
        if (const char* e = getenv("SPSP_PER_DEVICE")) { const long v = atol(e); if (v > 0) per_device = (size_t)v; }
        const int use = (int)std::max<size_t>(1, std::min<size_t>((size_t)visible, names.size() / per_device));
        for (int d = 0; d < use; ++d) devices.push_back(d);
    }
    // the progress lines of the reference (Comparator.cpp:56,69,364,414,503,509) are printed by the driver where the
    // reference prints them
    const int rc = spsp_compare_files_multi(devices.data(), (uint32_t)devices.size(), paths.data(), (uint32_t)paths.size(), n_query, (int)p,
                                            min_threshold, output_name.c_str(), query == "" ? 1 : 2, nullptr);
    if (rc != SPSP_OK) { cout << "Comparison failed: " << spsp_last_error() << endl; return 1; }
    return 0;
}
