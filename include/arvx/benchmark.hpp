// benchmark.hpp -- a timing sink (include/arvx/voxel_carving.hpp, setTimingSink) that keeps
// what the reference's Benchmark singleton keeps (src/Benchmark.h:23-151): per run a name,
// the model size and start/stop pairs for carving, colouring, post-processing, marching cubes
// and the whole run, printed as the reference's table (src/Benchmark.h:131-150: same header,
// same separators, milliseconds).  steady_clock instead of the reference's system_clock.
#ifndef ARVX_BENCHMARK_HPP
#define ARVX_BENCHMARK_HPP

#include <chrono>
#include <sstream>
#include <string>
#include <vector>

#include "arvx/voxel_carving.hpp"

namespace arvx {

class Benchmark {
   public:
    using Clock = std::chrono::steady_clock;
    struct Span {
        Clock::time_point start = Clock::now(), end = start;
        float ms() const { return std::chrono::duration<float, std::milli>(end - start).count(); }
        void log(bool begin) { (begin ? start : end) = Clock::now(); }
    };
    struct Run {
        std::string name;
        Vec4f size;  // x, y, z, voxel size
        Span stage[4], overall;
    };

    static Benchmark &GetInstance() {
        static Benchmark instance;
        return instance;
    }
    Benchmark(const Benchmark &) = delete;
    void operator=(const Benchmark &) = delete;

    void NextRun(const std::string &runName, Vec4f modelSize) {
        runs_.push_back(Run{runName, modelSize, {}, {}});
    }
    void LogCarving(bool start) { runs_.back().stage[kStageCarving].log(start); }
    void LogColoring(bool start) { runs_.back().stage[kStageColoring].log(start); }
    void LogPostProcessing(bool start) { runs_.back().stage[kStagePostProcessing].log(start); }
    void LogMarchingCubes(bool start) { runs_.back().stage[kStageMarchingCubes].log(start); }
    void LogOverall(bool start) { runs_.back().overall.log(start); }

    // route the library's stage brackets into this object
    void attach() {
        setTimingSink([this](Stage st, bool start) { runs_.back().stage[st].log(start); });
    }
    const std::vector<Run> &runs() const { return runs_; }

    std::string to_string() const {
        std::ostringstream ss;
        ss << std::endl << "Benchmark (all times in milliseconds)" << std::endl;
        ss << "Name\t\t\t\t" << "|  Model size (x,y,z, voxel size)\t" << "|  Carving time\t"
           << "|  Coloring time\t" << "|  Postprocessing time\t" << "|  Marching cubes time\t"
           << "|  Overall time" << std::endl;
        ss << std::string(177, '-') << std::endl;
        for (size_t i = 1; i < runs_.size(); i++) {  // run 0 is the placeholder, as in the reference
            const Run &r = runs_[i];
            ss << r.name << "|  " << r.size.x() << "x" << r.size.y() << "x" << r.size.z() << ", "
               << r.size(3) << "\t\t\t|  " << r.stage[kStageCarving].ms() << "\t|  "
               << r.stage[kStageColoring].ms() << "\t\t|  " << r.stage[kStagePostProcessing].ms()
               << "\t\t|  " << r.stage[kStageMarchingCubes].ms() << "\t\t|  " << r.overall.ms()
               << std::endl;
        }
        return ss.str();
    }

   private:
    // (a first run so that logging works before NextRun, as in the reference; and the library's
    // own stage brackets land here from the first GetInstance() on, as the reference's stages
    // log into its singleton themselves: src/VoxelCarving.cpp:62,70)
    Benchmark() {
        NextRun("dummy", Vec4f(0, 0, 0, 0));
        attach();
    }
    std::vector<Run> runs_;
};

}  // namespace arvx
#endif
