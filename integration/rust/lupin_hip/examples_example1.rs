// The reference's lupin_examples/src/example1.rs loop against this crate (move to examples/example1.rs in a Cargo
// workspace that also has a Cornell-box `SceneArrays`, e.g. from lupin_loader's `build_scene_cornell_box` CPU half).
// SOURCE ONLY -- see Cargo.toml.
use lupin_hip as lp;

fn render(device: &lp::Device, scene: &lp::Scene, camera_transform: lp::ffi::LupinMat3x4, camera_params: lp::CameraParams) -> Vec<u16> {
    let pathtrace_res = lp::build_pathtrace_resources(device, &lp::BakedPathtraceParams { with_runtime_checks: false, max_bounces: 8, samples_per_pixel: 5 });
    let mut output = lp::DoubleBufferedTexture::create(device, 1000, 1000);
    let num_accums = 200;
    for accum_idx in 0..num_accums {
        let desc = lp::PathtraceDesc {
            accum_params: Some(lp::AccumulationParams { prev_frame: output.back(), accum_counter: accum_idx }),
            tile_params: None, camera_params, camera_transform, force_software_bvh: true, advanced: Default::default(),
        };
        lp::pathtrace_scene(device, &pathtrace_res, scene, output.front(), lp::PathtraceType::Standard, &desc);
        output.flip();
    }
    output.flip();
    output.front().download()
}
