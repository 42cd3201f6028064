// imm3_sql -- the reference's SqlCli (engine/src/main/scala/immutabledb/SqlCli.scala:22-76) on the GPU path.
//   imm3_sql -q "select id, age from test_100 where (age > 18 and age < 30) limit 10" -d <dataDir> [--device n]
// Prints one `Row(...)` per line, like `println(it.next)` (SqlCli.scala:72).
//   --parse-only   print the parsed Query ADT and the planner's column order / leaves; no GPU needed.
#include <chrono>
#include <cstdio>
#include <cstring>
#include <iostream>

#include "../operators.hpp"
#include "../sql.hpp"

using namespace immutabledb;

static std::string showSelect(const SelectADT &s) {
    switch (s.kind) {
    case SelectADT::And: return "And(" + showSelect(*s.op1) + "," + showSelect(*s.op2) + ")";
    case SelectADT::Or: return "Or(" + showSelect(*s.op1) + "," + showSelect(*s.op2) + ")";
    case SelectADT::Select: return "Select(" + s.col + "," + s.cond.toString() + ")";
    default: return "NoSelect";
    }
}

static std::string showQuery(const Query &q) {
    std::string p;
    auto list = [](const std::vector<std::string> &v) { std::string s = "List("; for (size_t i = 0; i < v.size(); ++i) { if (i) s += ", "; s += v[i]; } return s + ")"; };
    if (q.project.kind == ProjectADT::Project) p = "Project(" + list(q.project.cols) + "," + std::to_string(q.project.limit) + ")";
    else {
        static const char *names[] = {"Sum", "Avg", "Min", "Max", "Count"};
        p = "ProjectAgg(List(";
        for (size_t i = 0; i < q.project.aggs.size(); ++i) { if (i) p += ", "; p += std::string(names[q.project.aggs[i].kind]) + "(" + q.project.aggs[i].col + ",None)"; }
        p += ")," + list(q.project.groupBy) + ")";
    }
    return "Query(" + q.table + "," + showSelect(*q.select) + "," + p + ")";
}

int main(int argc, char **argv) {
    std::string query, dataDir;
    int device = 0, repeat = 0;
    bool parseOnly = false;
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        if ((a == "-q" || a == "--query") && i + 1 < argc) query = argv[++i];
        else if ((a == "-d" || a == "--data-dir") && i + 1 < argc) dataDir = argv[++i];
        else if (a == "--device" && i + 1 < argc) device = std::atoi(argv[++i]);
        else if (a == "--cpu-count" && i + 1 < argc) ++i; // accepted for SqlCli compatibility; segments run on the GPU
        else if (a == "--parse-only") parseOnly = true;
        else if (a == "--repeat" && i + 1 < argc) repeat = std::atoi(argv[++i]); // re-run the query N times on the resident table, time to stderr
        else { std::fprintf(stderr, "Error parsing arguments: %s\n", a.c_str()); return 2; }
    }
    if (query.empty() || (dataDir.empty() && !parseOnly)) {
        std::fprintf(stderr, "Usage: imm3_sql -q <sql> -d <dataDir> [--device n] [--parse-only]\n");
        return 2;
    }
    try {
        const Query q = SQLParser::parseAll(query);
        if (parseOnly) {
            std::cout << showQuery(q) << "\n";
            if (!dataDir.empty()) {
                SegmentManager sm(dataDir);
                const Table &t = sm.getTable(q.table);
                std::cout << "usedColumns:";
                for (const auto &c : Engine::getColumns(q, t)) std::cout << " " << c.name;
                std::cout << "\nleaves:";
                for (const auto &l : Engine::resolveSelectOps(q)) std::cout << " " << l.col << ":" << l.cond.toString();
                std::cout << "\n";
            }
            return 0;
        }
        SegmentManager sm(dataDir);
        GpuSegmentManager gsm(sm, device);
        Engine engine(gsm);
        for (const Row &r : engine.execute(q)) std::cout << r.toString() << "\n";
        for (int k = 0; k < repeat; ++k) { // segments are resident now: this is the steady-state query time
            const auto t0 = std::chrono::steady_clock::now();
            const size_t n = engine.execute(q).size();
            const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            std::fprintf(stderr, "repeat %d: %.3f ms (%zu rows)\n", k, ms, n);
        }
    } catch (const std::exception &e) {
        std::cout << e.what() << "\n"; // res.fold(err => println(err), ...)
        return 1;
    }
    return 0;
}
