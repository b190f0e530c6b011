function scene = runDriverHip(sceneID, batch, itype)
%runDriverHip  Body shared by matlab/driverRedMaxBDF1.m and driverRedMaxBDF2.m (itype 1 / 2, the reference's numbering).
% Builds the scene with the reference's scenesRedMax, runs redmax.simLoopHip (all steps on the device) and hands the
% result to the reference's Scene.plotEnergies, which prints PASS / FAIL against Hexpected(itype).
scene = scenesRedMax(sceneID);
scene.init();
if batch
	[scene.drawHz, scene.computeH, scene.plotH] = deal(0, true, false);
else
	scene.test();      % the reference's finite-difference self tests run on its own MATLAB code, unchanged
	scene.draw();
end
fprintf('(%d) ''%s'' on HIP: %d steps of h=%g, nr=%d, nm=%d\n', sceneID, scene.name, scene.nsteps, scene.h, ...
	redmax.Scene.countR(), redmax.Scene.countM());
redmax.simLoopHip(scene, itype);
scene.plotEnergies(itype);
end
