// Drives the C++ host mirror the way the reference's own drivers do
// (apps/visual_test_correspondence_finder_projective_2d.cpp:71-97, apps/visual_test_aligner_2d.cpp:102-156):
//   host_mirror_driver fixed.bin moving.bin x y theta cols iterations
// reads two float32 [N,4] clouds, runs the finder at the given pose and the aligner from it, prints JSON.
#include <cstdio>
#include <cstdlib>
#include <lsm2d.hpp>

using namespace lsm2d_host;

static PointNormal2fVectorCloud read_cloud(const char* path) {
  FILE* f = fopen(path, "rb"); if (!f) { perror(path); exit(2); }
  fseek(f, 0, SEEK_END); long n = ftell(f) / (long) sizeof(PointNormal2f); fseek(f, 0, SEEK_SET);
  PointNormal2fVectorCloud c((size_t) n);
  if (n && fread(c.data(), sizeof(PointNormal2f), (size_t) n, f) != (size_t) n) exit(2);
  fclose(f); return c;
}

int main(int argc, char** argv) {
  if (argc < 8) { fprintf(stderr, "usage: %s fixed.bin moving.bin x y theta cols iterations\n", argv[0]); return 2; }
  try {
    Context ctx(0);
    PointNormal2fVectorCloud fixed = read_cloud(argv[1]), moving = read_cloud(argv[2]);
    Vector3f pose{{(float) atof(argv[3]), (float) atof(argv[4]), (float) atof(argv[5])}};
    const int cols = atoi(argv[6]), its = atoi(argv[7]);

    // reference error behaviour: compute() without inputs throws
    int threw = 0;
    { CorrespondenceFinderProjective2f empty(ctx); CorrespondenceVector cv; empty.setCorrespondences(&cv);
      try { empty.compute(); } catch (const std::runtime_error&) { threw = 1; } }

    std::shared_ptr<CorrespondenceFinderProjective2f> cf(new CorrespondenceFinderProjective2f(ctx));
    cf->param_projector->param_canvas_cols = cols; cf->param_projector->param_range_max = 30.f;
    cf->param_projector->param_angle_col_min = -(float) M_PI; cf->param_projector->param_angle_col_max = (float) M_PI;
    CorrespondenceVector correspondences;
    cf->setFixed(&fixed); cf->setMoving(&moving); cf->setLocalMapInSensor(pose); cf->setCorrespondences(&correspondences);
    cf->compute();

    AlignerSliceProcessorLaser2DPtr slice(new AlignerSliceProcessorLaser2D);
    slice->param_finder = cf; slice->param_min_num_correspondences = 10;
    MultiAligner2D::PropertyContainer fixed_props{{"points", &fixed}}, moving_props{{"points", &moving}};
    MultiAligner2D aligner(ctx);
    aligner.param_max_iterations = its; aligner.param_slice_processors.push_back(slice);
    aligner.setFixed(&fixed_props); aligner.setMoving(&moving_props); aligner.setMovingInFixed(pose);
    aligner.compute();

    printf("{\"threw_on_missing_inputs\": %d, \"n_pairs\": %zu, \"pairs\": [", threw, correspondences.size());
    for (size_t i = 0; i < correspondences.size(); ++i) printf("%s[%d,%d]", i ? "," : "", correspondences[i].fixed_idx, correspondences[i].moving_idx);
    const Vector3f& x = aligner.movingInFixed();
    printf("], \"status\": %d, \"pose\": [%.9g, %.9g, %.9g], \"iterations\": %zu, \"last_n_corr\": %d}\n", aligner.status(), x[0], x[1], x[2],
           aligner.iterationStats().size(), aligner.iterationStats().empty() ? 0 : aligner.iterationStats().back().n_correspondences);
  } catch (const std::exception& e) { fprintf(stderr, "error: %s\n", e.what()); return 1; }
  return 0;
}
