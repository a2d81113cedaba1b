// sub_sampler -- drop-in command line of the reference's sketcher
// (SubSampler.cpp:667-803) over libspsp.  Same flags, defaults, output names
// and stdout chatter; the scan itself runs on the GPU through the C-ABI.
#include <getopt.h>
#include <sys/stat.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <mutex>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

#include "../../include/spsp.h"

using namespace std;

// get_out_name (SubSampler.cpp:196-221): prefix + basename up to the first '.'
static string out_name(const string& path, const string& prefix) {
    size_t begin = 0;
    for (size_t i = 0; i < path.size(); ++i)
        if (path[i] == '/') begin = i + 1;
    string stem;
    for (size_t i = begin; i < path.size() && path[i] != '.'; ++i) stem.push_back(path[i]);
    return prefix + stem;
}

// intToString (utils.cpp:115-127)
static string with_commas(uint64_t n) {
    string s = to_string(n), out;
    for (size_t i = 0; i < s.size(); ++i) {
        out.push_back(s[i]);
        const size_t left = s.size() - 1 - i;
        if (left && left % 3 == 0) out.push_back(',');
    }
    return out;
}

static uint64_t file_size(const string& p) {
    struct stat st;
    return stat(p.c_str(), &st) == 0 ? (uint64_t)st.st_size : 0;
}

// print_stat (SubSampler.cpp:633-665), line for line.  total_kmer_number / total_superkmer_number cover ALL
// super-k-mers of the input: the library counts them in an extra pass when SPSP_SCAN_STATS is set (-v 1).
static void print_stat(const spsp_sketch_stats& s, uint32_t k, uint32_t m, const string& file) {
    if (s.selected_kmer_number == 0) { cout << "No kmer selected ***Crickets noise***" << endl; return; }
    cout << "I have seen " << with_commas(s.total_kmer_number) << " kmers and I selected " << with_commas(s.selected_kmer_number) << " kmers" << endl;
    cout << "After removing duplicate kmers, I selected " << with_commas(s.seen_kmers_at_reconstruction) << " kmers" << endl;
    cout << "This means a practical subsampling rate of " << (double)s.total_kmer_number / s.selected_kmer_number << " with duplicates" << endl;
    cout << "This means a practical subsampling rate of " << (double)s.total_kmer_number / s.seen_kmers_at_reconstruction << " without duplicates" << endl;
    cout << "I have seen " << with_commas(s.total_superkmer_number) << " superkmers and I selected " << with_commas(s.selected_superkmer_number) << " superkmers" << endl;
    cout << "After reconstruction and filtering with abundance, I have selected " << with_commas(s.seen_superkmers_at_reconstruction) << " superkmers" << endl;
    cout << "This means a practical subsampling rate of " << (double)s.total_superkmer_number / s.selected_superkmer_number << " with duplicates" << endl;
    cout << "This means a practical subsampling rate of " << (double)s.total_superkmer_number / s.seen_superkmers_at_reconstruction << " without duplicates" << endl;
    cout << "This means a mean superkmer size of " << (double)s.total_kmer_number / s.total_superkmer_number << " kmer per superkmer in the input" << endl;
    cout << "This means a mean superkmer size of " << (double)s.selected_kmer_number / s.selected_superkmer_number << " kmer per superkmer with duplicates" << endl;
    cout << "This means a mean superkmer size of " << (double)s.seen_kmers_at_reconstruction / s.seen_superkmers_at_reconstruction << " kmer per superkmer in the output" << endl;

    cout << "Actual output file size is " << with_commas(file_size(file) / 1000) << "KB" << endl;
    cout << "This mean " << ((double)file_size(file) * 8 / s.seen_kmers_at_reconstruction) << " bits per kmer" << endl;
    cout << "Minimizer number: " << with_commas(s.actual_minimizer_number) << " Skmer/minimizer:                    " << s.selected_superkmer_number / s.actual_minimizer_number << endl;
    cout << "Minimizer number: " << with_commas(s.actual_minimizer_number) << " Skmer/minimizer without duplicates: " << s.seen_superkmers_at_reconstruction / s.actual_minimizer_number << endl;
    cout << "Density is: " << (((double)s.selected_superkmer_number / s.nb_mmer_selected) * (k - m + 2)) << endl;
    cout << "Number of maximal skmer was:       " << with_commas(s.count_maximal_skmer) << endl;
    cout << "Actual number of maximal skmer is: " << with_commas(s.seen_max_superkmers_at_reconstruction) << endl;
    cout << "Proportion of max skmers:        " << ((double)s.count_maximal_skmer / s.selected_superkmer_number) * 100 << "% with duplicate kmers" << endl;
    cout << "Actual proportion of max skmers: " << ((double)s.seen_max_superkmers_at_reconstruction / s.seen_superkmers_at_reconstruction) * 100 << "%" << endl;
    cout << "\n" << endl;
}

int main(int argc, char** argv) {
    int ch;
    string input, inputfof, output("subsampled_");
    unsigned k = 31, m1 = 11, c = 8, abundance = 1;
    double s = 1000;
    bool verbose = true;
    while ((ch = getopt(argc, argv, "hdg:q:k:m:n:s:t:b:e:f:i:p:v:x:a:")) != -1) {
        switch (ch) {
            case 'i': input = optarg; break;
            case 'f': inputfof = optarg; break;
            case 'k': k = stoi(optarg); break;
            case 'm': m1 = stoi(optarg); break;
            case 't': c = stoi(optarg); break;
            case 's': s = stof(optarg); break;  // float, as the reference (SubSampler.cpp:699)
            case 'p': output = optarg; break;
            case 'v': verbose = stoi(optarg); break;
            case 'x': break;  // stored but unused by the reference's live code
            case 'a': abundance = stoi(optarg); break;
        }
    }
    if (input == "" && inputfof == "") {
        cout << "Core arguments:" << endl
             << "	-i Input file" << endl
             << "	-f Input file of file" << endl
             << "	-p Output prefix (subsampled)" << endl
             << "	-k Kmer size used  (31) " << endl
             << "	-s Subsampling used  (1000) " << endl
             << "	-t Threads used  (8) " << endl
             << "	-m Minimizer size used  (11, max value is 15) " << endl
             << "	-v Verbose level (1) " << endl
             << "	-a Abundance min (2) " << endl
             << "	-3/2/1 respectively Max skmers + any sized skmers + cursed skmers OR Max skmers and any sized skmers OR max skmers only. (default 3) " << endl;
        return 0;
    }
    if (m1 % 2 == 0) { cout << "Minimizer size must be odd" << endl; m1++; }
    if (k % 2 == 0) { cout << "Kmer size must be odd" << endl; k++; }
    if (m1 > 15) { cout << "Minimizer size can't be greater than 15." << endl; m1 = 15; }
    cout << " I use k=" << k << " m=" << m1 << " s=" << s << endl;
    cout << "Maximal super kmer are of length " << 2 * k - m1 << " or " << k - m1 + 1 << " kmers" << endl;
    if (k > 63 || m1 > k) { cout << "k must satisfy m <= k <= 63" << endl; return 0; }

    spsp_params P;
    P.k = k; P.m = m1; P.abundance = abundance;
    P.flags = verbose ? SPSP_SCAN_STATS : SPSP_SCAN_DEFAULT;   // print_stat needs the count of ALL super-k-mers
    P.threshold = spsp_threshold_host(k, m1, s);

    // The reference's `#pragma omp parallel num_threads(c)` loop (SubSampler.cpp:771-793) lives in the library
    // (spsp_sketch_files): c workers, one context (= one HIP stream) each, files taken off the list in order.  The
    // callbacks print what the reference prints where it prints it: the file name (and the line of the output list) when
    // a file is taken, print_stat when it is done.
    struct Run {
        vector<string> files, outs;
        ofstream* out_fof;
        bool verbose; uint32_t k, m;
        bool gpu_ok = true;
    } run;
    run.verbose = verbose; run.k = k; run.m = m1; run.out_fof = nullptr;
    auto cb = [](void* user, uint32_t i, int phase, int rc, const spsp_sketch_stats* st, const char* err) {
        Run& r = *(Run*)user;
        // the library serialises the phase-0 calls among themselves and the phase-1 calls among themselves; stdout is one
        // stream: a file-name line must not land inside another file's print_stat block (critical(cout), SubSampler.cpp:782,791)
        static std::mutex out_mutex;
        std::lock_guard<std::mutex> lock(out_mutex);
        if (phase == 0) {
            if (r.out_fof) {                                  // :784-785 (file-of-files mode only)
                cout << r.files[i] << endl;
                *r.out_fof << r.outs[i] << "\n";
            }
            return;
        }
        if (rc != SPSP_OK) { cout << "Can't process file: " << r.files[i] << " (" << (err ? err : "") << ")" << endl; return; }
        if (r.verbose) print_stat(*st, r.k, r.m, r.outs[i]);
    };
    auto sketch_all = [&](unsigned threads) {
        vector<const char*> in, out;
        for (auto& f : run.files) in.push_back(f.c_str());
        for (auto& f : run.outs) out.push_back(f.c_str());
        // devices: SPSP_DEVICES="0,1,..." names them; else every visible GPU once there are enough files to deal (sketching
        // shards by genome: spsp_sketch_files_multi deals the batches, nothing is exchanged)
        vector<int> devices;
        if (const char* e = getenv("SPSP_DEVICES")) {
            istringstream is(e);
            string tok;
            while (getline(is, tok, ',')) if (!tok.empty()) devices.push_back(atoi(tok.c_str()));
        }
        if (devices.empty()) {
            const int visible = spsp_device_count();
            long per_device = 16;                               // files per device from which dealing pays; SPSP_PER_DEVICE=<n> overrides
            if (const char* e = getenv("SPSP_PER_DEVICE")) { const long v = atol(e); if (v > 0) per_device = v; }
            const int use = std::max(1, std::min(visible, (int)((long)in.size() / per_device)));
            for (int d = 0; d < use; ++d) devices.push_back(d);
        }
        const int rc = spsp_sketch_files_multi(devices.data(), (uint32_t)devices.size(), &P, s, in.data(), out.data(), (uint32_t)in.size(), threads, cb, &run, nullptr);
        if (rc == SPSP_ERR_NO_DEVICE || rc == SPSP_ERR_HIP) { cout << "GPU unavailable: " << spsp_last_error() << endl; run.gpu_ok = false; }
    };

    if (input != "") {
        run.files.push_back(input);
        run.outs.push_back(out_name(input, output) + ".gz");  // written to the CWD, like the reference
        sketch_all(1);
        return run.gpu_ok ? 0 : 1;
    }
    uint8_t* fof_data = nullptr; uint64_t fof_len = 0;
    if (spsp_read_file_host(inputfof.c_str(), &fof_data, &fof_len) != SPSP_OK) { cout << "Can't open file of file " << inputfof << endl; return 0; }
    {
        istringstream is(string((const char*)fof_data, fof_len));
        string line;
        while (getline(is, line))
            if (line.size() > 3) { run.files.push_back(line); run.outs.push_back(out_name(line, output) + ".gz"); }
        spsp_free(fof_data);
    }
    ofstream out_fof(out_name(inputfof, output) + ".txt");
    run.out_fof = &out_fof;
    if (c == 0) c = 1;
    sketch_all(c);
    out_fof.close();
    const bool gpu_ok = run.gpu_ok;
    return gpu_ok ? 0 : 1;
}
