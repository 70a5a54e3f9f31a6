// histogram.h — the text histogram the reference logs its residual distributions with (src/common/histogram.h:23-30,
// histogram.cc:27-76: a Cartographer utility): Count / Min / Max / Mean in FLOAT arithmetic, then `buckets` equal-width
// buckets with a 20-character bar, count and running total.  Used by LidarOdometry's optional residual log
// (PrintSurfelResiduals / PrintImuResiduals, lidar_odometry.cc:56-94).  Same numbers and the same layout; std::snprintf
// stands in for absl::StrCat / StrAppendFormat ("%g" for the values StrCat prints with six significant digits).
#pragma once
#include <algorithm>
#include <cstdio>
#include <string>
#include <vector>

class Histogram {
 public:
  void Add(double value) { values_.push_back(value); }
  size_t size() const { return values_.size(); }

  std::string ToString(int buckets) const {
    if (buckets < 1) buckets = 1;  // (CHECK_GE(buckets, 1) in the reference)
    if (values_.empty()) return "Count: 0";
    const float min = (float)*std::min_element(values_.begin(), values_.end());
    const float max = (float)*std::max_element(values_.begin(), values_.end());
    float sum = 0.f;  // std::accumulate(..., 0.f): a float accumulator
    for (const double v : values_) sum = (float)(sum + v);
    const float mean = sum / values_.size();
    char buf[256];
    std::snprintf(buf, sizeof(buf), "Count: %zu  Min: %g  Max: %g  Mean: %g", values_.size(), (double)min, (double)max, (double)mean);
    std::string result = buf;
    if (min == max) return result;
    float lower_bound = min;
    int total_count = 0;
    for (int i = 0; i != buckets; ++i) {
      const float upper_bound = (i + 1 == buckets) ? max : (max * (i + 1) / buckets + min * (buckets - i - 1) / buckets);
      int count = 0;
      for (const float value : values_)  // (the reference narrows every value to float here too)
        if (lower_bound <= value && (i + 1 == buckets ? value <= upper_bound : value < upper_bound)) ++count;
      total_count += count;
      std::snprintf(buf, sizeof(buf), "\n[%f, %f%c", (double)lower_bound, (double)upper_bound, i + 1 == buckets ? ']' : ')');
      result += buf;
      constexpr int kMaxBarChars = 20;
      const int bar = (int)((count * (size_t)kMaxBarChars + values_.size() / 2) / values_.size());
      result += "\t";
      for (int c = 0; c != kMaxBarChars; ++c) result += (c < (kMaxBarChars - bar)) ? " " : "#";
      std::snprintf(buf, sizeof(buf), "\tCount: %d (%g%%)\tTotal: %d (%g%%)", count, (double)(count * 1e2f / values_.size()), total_count,
                    (double)(total_count * 1e2f / values_.size()));
      result += buf;
      lower_bound = upper_bound;
    }
    return result;
  }

 private:
  std::vector<double> values_;
};
