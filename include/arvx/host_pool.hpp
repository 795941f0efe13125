// host_pool.hpp -- recycled, page-locked host memory for the big arrays that cross PCIe.
//
// The drop-in layer hands results to the caller in std::vectors, as the reference does
// (SimpleMesh::m_vertices / m_triangles, src/MarchingCubes.h:33-92; the colour lists behind
// Model::get).  At 512^3 a mesh is 77 MB and a fresh std::vector of that size costs more than
// everything the GPU does for it: first-touch page faults (~20 ms) and a copy through the
// runtime's staging buffers because the pages are pageable.  PooledAllocator keeps the blocks of
// destroyed vectors (process-wide, bounded) and page-locks them once (arvx_host_register), so
// that the next mesh / colour list downloads straight into mapped, pinned memory at PCIe rate;
// and it default-constructs trivially copyable elements without writing them (resize() before a
// download must not memset 77 MB).
#ifndef ARVX_HOST_POOL_HPP
#define ARVX_HOST_POOL_HPP

#include <cstdlib>
#include <mutex>
#include <new>
#include <type_traits>
#include <utility>
#include <vector>

#include "arvx/arvx.h"

namespace arvx {
namespace detail {

class HostPool {
   public:
    static constexpr size_t kSmall = 256 * 1024;     // below this: plain malloc
    // idle blocks kept, in total (page-locked memory stays locked while it idles): 1 GiB unless
    // the program says otherwise -- HostPool::instance().set_keep_bytes(...)
    size_t keep_bytes_ = 1ull << 30;
    void set_keep_bytes(size_t bytes) {
        std::lock_guard<std::mutex> lock(m_);
        keep_bytes_ = bytes;
    }
    static HostPool &instance() {
        static HostPool *pool = new HostPool();  // never destroyed: no HIP calls at exit
        return *pool;
    }
    void *take(size_t bytes) {
        if (bytes < kSmall) {
            void *p = std::malloc(bytes ? bytes : 1);
            if (!p) throw std::bad_alloc();
            return p;
        }
        {
            std::lock_guard<std::mutex> lock(m_);
            int best = -1;
            for (int i = 0; i < (int)blocks_.size(); ++i)
                if (!blocks_[i].used && blocks_[i].cap >= bytes && blocks_[i].cap <= 2 * bytes + kSmall &&
                    (best < 0 || blocks_[i].cap < blocks_[best].cap))
                    best = i;
            if (best >= 0) {
                blocks_[best].used = true;
                idle_ -= blocks_[best].cap;
                return blocks_[best].p;
            }
        }
        // whole 2 MiB pages and a quarter of slack: the next, slightly larger array still fits
        const size_t cap = (bytes + bytes / 4 + (2u << 20) - 1) & ~(size_t)((2u << 20) - 1);
        void *p = nullptr;
        if (posix_memalign(&p, 2u << 20, cap) != 0 || !p) throw std::bad_alloc();
        const bool pinned = arvx_host_register(p, cap) == ARVX_OK;  // best effort
        std::lock_guard<std::mutex> lock(m_);
        blocks_.push_back(Block{p, cap, true, pinned});
        return p;
    }
    void give(void *p, size_t bytes) {
        if (bytes < kSmall) {
            std::free(p);
            return;
        }
        std::lock_guard<std::mutex> lock(m_);
        for (size_t i = 0; i < blocks_.size(); ++i)
            if (blocks_[i].p == p) {
                if (idle_ + blocks_[i].cap > keep_bytes_) {
                    drop(i);
                } else {
                    blocks_[i].used = false;
                    idle_ += blocks_[i].cap;
                }
                return;
            }
        std::free(p);  // (not ours: cannot happen)
    }
    // give the idle blocks back to the system (e.g. before a long phase without meshes)
    void trim() {
        std::lock_guard<std::mutex> lock(m_);
        for (size_t i = blocks_.size(); i-- > 0;)
            if (!blocks_[i].used) {
                idle_ -= blocks_[i].cap;
                drop(i);
            }
    }

   private:
    struct Block {
        void *p;
        size_t cap;
        bool used, pinned;
    };
    void drop(size_t i) {
        if (blocks_[i].pinned) (void)arvx_host_unregister(blocks_[i].p);
        std::free(blocks_[i].p);
        blocks_.erase(blocks_.begin() + (long)i);
    }
    std::mutex m_;
    std::vector<Block> blocks_;
    size_t idle_ = 0;
};

template <class T>
struct PooledAllocator {
    using value_type = T;
    PooledAllocator() = default;
    template <class U>
    PooledAllocator(const PooledAllocator<U> &) {}
    T *allocate(size_t n) { return static_cast<T *>(HostPool::instance().take(n * sizeof(T))); }
    void deallocate(T *p, size_t n) { HostPool::instance().give(p, n * sizeof(T)); }
    // value-initialisation of trivially copyable elements is skipped: every user fills the
    // array right after resize()
    template <class U, class... A>
    void construct(U *p, A &&...a) {
        if constexpr (sizeof...(A) == 0 && std::is_trivially_copyable<U>::value) {
            (void)p;
        } else {
            ::new ((void *)p) U(std::forward<A>(a)...);
        }
    }
    template <class U>
    bool operator==(const PooledAllocator<U> &) const { return true; }
    template <class U>
    bool operator!=(const PooledAllocator<U> &) const { return false; }
};

}  // namespace detail

// std::vector on recycled page-locked memory; elements are NOT zeroed by resize()
template <class T>
using HostVector = std::vector<T, detail::PooledAllocator<T>>;

// return the idle blocks of the pool to the system
inline void trimHostPool() { detail::HostPool::instance().trim(); }

}  // namespace arvx
#endif
