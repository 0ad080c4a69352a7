"""Host-side callers of the hot path, restated so that smoke(), bench.py and the tests can
drive the HIP operators the way the reference does when /root/reference is not available
(it is not shipped to the GPU box):

  options.default_opt()      the `opt` namespace of main.py after `-O`            (main.py:19-287)
  renderer.NeRFRenderer      run_cuda / update_extra_state / render               (nerf/renderer.py)
  network_grid.NeRFNetwork   hash-grid field + background MLP                     (nerf/network_grid.py)
  optim.Adan                 the optimiser `-O` training uses                     (optimizer.py, main.py:368)
  guidance.*                 SDS loss glue with pluggable noise predictors        (guidance/sd_utils.py:86-163)
  trainer.TrainStep          one iteration of Trainer.train_one_epoch             (nerf/utils.py:439-717,1032-1070)

With the reference checkout present, its own main.py / nerf/ run unchanged against the operator
packages; these modules are the same control flow without the reference's
unrelated imports.
"""
