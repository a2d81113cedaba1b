// sortCSV -- drop-in for the reference's third program (sort_csv.cpp:115-122):
//   sortCSV <jaccard.csv.gz> <out.csv> <original file of files>
// Host only (no GPU work): gunzip/plain autodetect on the input, plain text out.
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>

#include "../../include/spsp.h"

int main(int argc, char** argv) {
    if (argc < 4) {
        std::cout << "Need input, output filename and original fof" << std::endl;   // sort_csv.cpp:116-119
        return 0;
    }
    uint8_t *csv = nullptr, *fof = nullptr;
    uint64_t csv_len = 0, fof_len = 0;
    if (spsp_read_file_host(argv[1], &csv, &csv_len) || spsp_read_file_host(argv[3], &fof, &fof_len)) {
        std::cout << "cant open file" << std::endl;                                  // sort_csv.cpp:32-35
        std::cerr << spsp_last_error() << std::endl;
        return 0;
    }
    char* text = nullptr;
    uint64_t len = 0;
    const int rc = spsp_sort_csv_host((const char*)csv, csv_len, (const char*)fof, fof_len, &text, &len);
    spsp_free(csv);
    spsp_free(fof);
    if (rc) {
        std::cerr << "sortCSV: " << spsp_last_error() << std::endl;
        return 1;
    }
    std::ofstream out(argv[2], std::ios::binary);
    out.write(text, (std::streamsize)len);
    spsp_free(text);
    if (!out) { std::cerr << "sortCSV: cannot write " << argv[2] << std::endl; return 1; }
    std::cout << "The end" << std::endl;                                             // sort_csv.cpp:110
    return 0;
}
