// Power sampling around the kernel: the role of the reference's PowerMeter (absent submodule
// `powermeter`, a Corsair-PSU reader) as host/RunHardware.cpp:156-172 uses it -- construct with a
// sampling period, Start() before the launch, Stop() after it, average the samples.  Here the
// sensor is the GPU's own: ROCm SMI (librocm_smi64.so, located at run time like the BLAS because
// the host binaries must still start on a box without it) or, failing that, the amdgpu hwmon file
// /sys/class/drm/card<N>/device/hwmon/hwmon*/power1_average|power1_input (microwatts).
// The sensor is looked up by the PCI address of the HIP device (mm_device_pci_bus_id): ROCm SMI and sysfs enumerate
// physical devices and ignore HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES, so the HIP ordinal is not their index
// (ADVICE r2).  A background thread samples every `period_ms` while the kernel runs (the verdict's point: a
// sample taken after the process has exited measures an idle chip).
#pragma once
#include <dlfcn.h>

#include <atomic>
#include <cctype>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <string>
#include <thread>
#include <vector>

#include <dirent.h>

namespace mmhost {

class PowerMeter {
 public:
  // `pci_bus_id`: "dddd:bb:dd.f" of the GPU that runs the kernel; empty = the device'th sensor in enumeration order
  PowerMeter(int device, int period_ms, const std::string &pci_bus_id = std::string())
      : device_(device), period_ms_(period_ms > 0 ? period_ms : 1), bdf_(pci_bus_id) {
    for (char &c : bdf_) c = (char)std::tolower((unsigned char)c);
    Open();
  }
  ~PowerMeter() {
    Stop();
    if (shut_down_) shut_down_();
  }
  PowerMeter(const PowerMeter &) = delete;
  PowerMeter &operator=(const PowerMeter &) = delete;

  void Start() {
    samples_.clear();
    stop_.store(false);
    t0_ = std::chrono::steady_clock::now();
    thread_ = std::thread([this] {
      while (!stop_.load(std::memory_order_relaxed)) {
        double w;
        if (ReadWatts(&w)) samples_.push_back(w);
        std::this_thread::sleep_for(std::chrono::milliseconds(period_ms_));
      }
    });
  }
  void Stop() {
    if (!thread_.joinable()) return;
    stop_.store(true);
    thread_.join();
  }
  double Elapsed() const { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0_).count(); }
  size_t Samples() const { return samples_.size(); }
  double Average() const {
    if (samples_.empty()) return 0.0;
    double s = 0;
    for (double w : samples_) s += w;
    return s / samples_.size();
  }
  const std::string &Source() const { return source_; }

 private:
  using init_t = int (*)(uint64_t);
  using shut_t = int (*)();
  using power_t = int (*)(uint32_t, uint64_t *, int *);       // rsmi_dev_power_get(dv_ind, &uW, &type)
  using ave_t = int (*)(uint32_t, uint32_t, uint64_t *);      // rsmi_dev_power_ave_get(dv_ind, sensor, &uW)
  using count_t = int (*)(uint32_t *);                         // rsmi_num_monitor_devices(&n)
  using pci_t = int (*)(uint32_t, uint64_t *);                 // rsmi_dev_pci_id_get(dv_ind, &bdfid)

  // ROCm SMI's BDFID: domain << 32 | bus << 8 | device << 3 | function
  static bool ParseBdf(const std::string &s, uint64_t *id) {
    unsigned dom = 0, bus = 0, dev = 0, fn = 0;
    if (std::sscanf(s.c_str(), "%x:%x:%x.%x", &dom, &bus, &dev, &fn) != 4) return false;
    *id = ((uint64_t)dom << 32) | ((uint64_t)bus << 8) | ((uint64_t)dev << 3) | fn;
    return true;
  }

  void Open() {
    for (const char *lib : {"librocm_smi64.so", "/opt/rocm/lib/librocm_smi64.so", "librocm_smi64.so.1"}) {
      void *h = dlopen(lib, RTLD_NOW | RTLD_LOCAL);
      if (!h) continue;
      auto init = (init_t)dlsym(h, "rsmi_init");
      power_ = (power_t)dlsym(h, "rsmi_dev_power_get");
      ave_ = (ave_t)dlsym(h, "rsmi_dev_power_ave_get");
      if (init && (power_ || ave_) && init(0) == 0) {
        shut_down_ = (shut_t)dlsym(h, "rsmi_shut_down");
        uint64_t want = 0;
        auto count = (count_t)dlsym(h, "rsmi_num_monitor_devices");
        auto pci = (pci_t)dlsym(h, "rsmi_dev_pci_id_get");
        if (!bdf_.empty() && ParseBdf(bdf_, &want) && count && pci) {   // the SMI index of THIS GPU
          uint32_t n = 0;
          if (count(&n) == 0)
            for (uint32_t i = 0; i < n; ++i) {
              uint64_t id = 0;
              if (pci(i, &id) == 0 && (id & 0xffffffff0000ffffull) == (want & 0xffffffff0000ffffull)) { device_ = (int)i; break; }
            }
        }
        double w;
        if (ReadWatts(&w)) {
          source_ = std::string("ROCm SMI ") + (used_ave_ ? "rsmi_dev_power_ave_get" : "rsmi_dev_power_get");
          return;
        }
        if (shut_down_) shut_down_();
        shut_down_ = nullptr;
      }
      power_ = nullptr;
      ave_ = nullptr;
    }
    // hwmon fallback: the sensor under this GPU's PCI device, else the device'th card that has a power sensor
    if (!bdf_.empty()) {
      const std::string base = "/sys/bus/pci/devices/" + bdf_ + "/hwmon";
      if (DIR *d = opendir(base.c_str())) {
        while (dirent *e = readdir(d)) {
          if (e->d_name[0] == '.') continue;
          for (const char *leaf : {"power1_average", "power1_input"}) {
            const std::string path = base + "/" + e->d_name + "/" + leaf;
            if (hwmon_path_.empty() && std::ifstream(path).good()) hwmon_path_ = path;
          }
        }
        closedir(d);
      }
    }
    int seen = 0;
    for (int card = 0; card < 64 && hwmon_path_.empty(); ++card) {
      const std::string base = "/sys/class/drm/card" + std::to_string(card) + "/device/hwmon";
      DIR *d = opendir(base.c_str());
      if (!d) continue;
      while (dirent *e = readdir(d)) {
        if (e->d_name[0] == '.') continue;
        for (const char *leaf : {"power1_average", "power1_input"}) {
          const std::string path = base + "/" + e->d_name + "/" + leaf;
          if (std::ifstream(path).good()) {
            if (seen++ == device_) hwmon_path_ = path;
            break;
          }
        }
        if (!hwmon_path_.empty()) break;
      }
      closedir(d);
    }
    source_ = hwmon_path_.empty() ? "no power sensor found" : hwmon_path_;
  }

  bool ReadWatts(double *watts) {
    uint64_t uw = 0;
    int type = 0;
    if (power_ && power_((uint32_t)device_, &uw, &type) == 0 && uw) {
      *watts = 1e-6 * (double)uw;
      return true;
    }
    if (ave_ && ave_((uint32_t)device_, 0, &uw) == 0 && uw) {
      used_ave_ = true;
      *watts = 1e-6 * (double)uw;
      return true;
    }
    if (!hwmon_path_.empty()) {
      std::ifstream f(hwmon_path_);
      if (f >> uw) {
        *watts = 1e-6 * (double)uw;
        return true;
      }
    }
    return false;
  }

  int device_, period_ms_;
  std::string bdf_;
  power_t power_ = nullptr;
  ave_t ave_ = nullptr;
  shut_t shut_down_ = nullptr;
  bool used_ave_ = false;
  std::string hwmon_path_, source_;
  std::vector<double> samples_;
  std::atomic<bool> stop_{true};
  std::thread thread_;
  std::chrono::steady_clock::time_point t0_ = std::chrono::steady_clock::now();
};

}  // namespace mmhost
