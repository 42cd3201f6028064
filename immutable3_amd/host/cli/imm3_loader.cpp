// imm3_loader -- the reference's LoaderCli (loader/src/main/scala/immutabledb/loader/LoaderCli.scala:27-155): CSV ->
// table directory in the reference's on-disk format.  Host-only (no GPU).
//   imm3_loader -t test_100 -c id:DENSE_INT,state:DENSE_STRING:size=2,age:DENSE_TINYINT -d <dataDir> -i data.csv
//               [--block-size 1024] [--segment-size 100]
#include <cstdio>
#include <iostream>

#include "../storage.hpp"

using namespace immutabledb;

static Column parseCol(const std::string &arg) { // parseCol / parseColOptions, LoaderCli.scala:66-81
    const std::vector<std::string> parts = split(arg, ':');
    if (parts.size() < 2) throw Exception("bad column definition: " + arg);
    std::vector<std::pair<std::string, std::string>> opts;
    if (parts.size() == 3)
        for (const auto &kv : split(parts[2], ';')) {
            const std::vector<std::string> xs = split(kv, '=');
            opts.emplace_back(xs.front(), xs.back());
        }
    if (parts[1] == "DENSE_INT") return Column::make(parts[0], CodecType::DENSE_INT);
    if (parts[1] == "PFOR_INT") return Column::make(parts[0], CodecType::PFOR_INT); // not offered by LoaderCli.scala:118-122; SegmentWriter handles it
    if (parts[1] == "SNAPPY_INT") return Column::make(parts[0], CodecType::SNAPPY_INT);             // extension codecs (schema.hpp)
    if (parts[1] == "SNAPPY_TINYINT") return Column::make(parts[0], CodecType::SNAPPY_TINYINT);
    if (parts[1] == "SNAPPY_STRING") return Column::make(parts[0], CodecType::SNAPPY_STRING, opts);
    if (parts[1] == "DENSE_TINYINT") return Column::make(parts[0], CodecType::DENSE_TINYINT);
    if (parts[1] == "DENSE_STRING") return Column::make(parts[0], CodecType::DENSE_STRING, opts);
    throw Exception("MatchError: " + parts[1]); // LoaderCli.scala:118-122 has no other case
}

int main(int argc, char **argv) {
    std::string table, cols, dataDir, input;
    int blockSize = 1024, segSize = 100; // DevEnv defaults (core/.../env.scala:24-32)
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        auto need = [&](const char *what) -> std::string { if (i + 1 >= argc) { std::fprintf(stderr, "missing value for %s\n", what); std::exit(2); } return argv[++i]; };
        if (a == "-t" || a == "--table-name") table = need("-t");
        else if (a == "-c" || a == "--cols") cols = need("-c");
        else if (a == "-d" || a == "--data-dir") dataDir = need("-d");
        else if (a == "-i" || a == "--input-csv") input = need("-i");
        else if (a == "--block-size") blockSize = std::atoi(need("--block-size").c_str());
        else if (a == "--segment-size") segSize = std::atoi(need("--segment-size").c_str());
        else { std::fprintf(stderr, "Error parsing arguments: %s\n", a.c_str()); return 2; }
    }
    if (table.empty() || cols.empty() || dataDir.empty() || input.empty()) {
        std::fprintf(stderr, "Usage: imm3_loader -t <table> -c <COL_DEF>,<COL_DEF>,... -d <dataDir> -i <csv> [--block-size n] [--segment-size n]\n");
        return 2;
    }
    try {
        Table t;
        t.name = table;
        t.blockSize = blockSize;
        for (const auto &c : split(cols, ',')) t.columns.push_back(parseCol(c));
        loadCsv(dataDir, t, input, segSize);
    } catch (const std::exception &e) {
        std::cerr << e.what() << "\n";
        return 1;
    }
    return 0;
}
